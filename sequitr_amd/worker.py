"""One job = one worker process: the boundary the MI355X engine sits behind.

What the reference's worker.py defines and this module honours (line numbers in sequitr/worker.py):
  * a job is an INI file ``<name>.job`` with a single ``[job]`` section; required keys ID, user,
    priority, device, module, func, params; optional options, lib_path (49-60, 86-107);
    params / options are Python literals (100-102);
  * running a job means ``import module; module.func(params, options)`` with the output folder
    injected as ``params['output']``; whatever func returns is dropped and whatever it raises ends up
    in the log, not in the caller (195-215 with serverlogs.py:113-127);
  * a finished job file is renamed ``<name>.job.complete`` (218-239);
  * ``python worker.py --job F --out D`` is the process entry (252-316) -- here
    ``python -m sequitr_amd.worker``.

Added for a multi-GPU node: a ``device = GPU`` job lands on the HIP device named by ``options['gpu']``
or, failing that, LOCAL_RANK (the server sets it; see jobs._resolve_device).  The reference checked
``device`` and then never used it.
"""
import argparse
import configparser
import importlib
import logging
import os
import sys
from ast import literal_eval

from . import serverlogs
from .utils import check_and_makedir

_DEVICES = ('CPU', 'GPU')
_HEADER = (('ID', 'ID'), ('owner', 'user'), ('priority', 'priority'), ('device', 'device'))   # ctor arg <- INI key


def _read_only(name):
    return property(lambda self: getattr(self, '_' + name))


class JobWrapper(object):
    """A parsed job: identity (ID / owner / priority / device / filename), the call it stands for
    (_module._func(_params, _options)) and its completion flag."""

    ID = _read_only('ID')
    owner = _read_only('owner')
    priority = _read_only('priority')
    device = _read_only('device')
    filename = _read_only('filename')

    def __init__(self, ID=None, filename=None, owner='root', priority=99, device='CPU'):
        if device not in _DEVICES:
            logging.warning('Device {0:s} not recognised'.format(str(device)))
            raise ValueError
        self._ID, self._filename, self._owner = ID, filename, owner
        self._priority, self._device = priority, device
        self._module = self._func = self._lib_path = None
        self._params, self._options = {}, {}
        self._complete = False

    # -- where the job writes ---------------------------------------------------------------------
    @property
    def job_output(self):
        return self._job_output

    @job_output.setter
    def job_output(self, folder):
        if not isinstance(folder, str):
            raise TypeError('Output folder must be a string')
        check_and_makedir(folder)
        self._job_output = folder

    # -- done? --------------------------------------------------------------------------------------
    @property
    def complete(self):
        return self._complete

    @complete.setter
    def complete(self, flag=False):
        if flag:
            try:
                os.rename(self._filename, self._filename + '.complete')
            except OSError:
                raise OSError('Failed to set job flags to complete')
            self._complete = True

    # -- run ------------------------------------------------------------------------------------------
    @serverlogs.exception_logger
    def __call__(self):
        if self._complete:
            return
        if self._lib_path:
            sys.path.append(self._lib_path)
        target = getattr(importlib.import_module(self._module), self._func)
        self._params['output'] = self.job_output
        self._params.setdefault('device', self._device)        # the job function sees 'CPU' | 'GPU'
        target(self._params, self._options)                    # return value dropped on purpose

    @staticmethod
    def load(filename, header_only=False):
        return parse_job_file(filename, header_only=header_only)


def parse_job_file(filename, header_only=False, quiet=False):
    """``<name>.job`` -> JobWrapper (None, with the reason logged, when the file is unusable).
    header_only stops after identity + module/func: enough for a server to queue the job.
    quiet: return None without logging (a server re-polling a file it has already reported once)."""
    if quiet:
        try:
            return _parse_job_file(filename, header_only)
        except Exception:                                      # noqa: BLE001 -- reported by the first, loud poll
            return None
    return _parse_job_file_logged(filename, header_only)


def _parse_job_file(filename, header_only=False):
    if not isinstance(filename, str):
        raise Exception("Job filename is not correctly formed")
    if not filename.endswith('.job'):
        raise IOError('Job {0:s} does not have .job file extenstion'.format(filename))
    ini = configparser.ConfigParser()
    ini.optionxform = str                                      # 'ID' must stay upper case
    ini.read(filename)

    def field(key, required=True):
        for k in (key, key.lower()):
            if ini.has_option('job', k):
                return ini.get('job', k)
        if required:
            raise configparser.NoOptionError(key, 'job')
        return None

    job = JobWrapper(filename=filename, **{arg: field(key) for arg, key in _HEADER})
    job._module, job._func = field('module'), field('func')
    if not header_only:
        job._params = literal_eval(field('params'))
        opts = field('options', required=False)
        if opts is not None:
            job._options = literal_eval(opts)
        lib_path = field('lib_path', required=False)
        if lib_path and os.path.exists(lib_path):
            job._lib_path = lib_path
    return job


@serverlogs.exception_logger
def _parse_job_file_logged(filename, header_only=False):
    return _parse_job_file(filename, header_only)


_parse_job_file_logged.__wrapped__.__name__ = 'parse_job_file'     # the name the reference's log line carries


def worker(args, log=True):
    """Process body: args.job (job file) and args.out (output folder) come from argparse -- the
    reference refuses anything else, so does this."""
    if not isinstance(args, argparse.Namespace):
        raise TypeError('Args must be of type argsparse.Namespace. The worker'
                        ' function cannot be called directly.')
    missing = [a for a in ('out', 'job') if a not in args]
    if missing:
        raise AttributeError('Could not find .out or .job in args.')
    check_and_makedir(args.out)
    if log:
        serverlogs.setup_logging(args.out, log_name='worker_process')
        logging.getLogger('worker_process').info(args.job)
    failures = serverlogs.failure_count()
    try:
        job = JobWrapper.load(args.job)
        if job is not None:
            job.job_output = args.out
            job()
    finally:
        if log:
            serverlogs.shutdown_logging('worker_process')
    # exceptions are logged and swallowed as in the reference (serverlogs.py:113-127), but the outcome is kept:
    # False when the job file could not be parsed or the job function raised
    return job is not None and serverlogs.failure_count() == failures


def main(argv=None):
    cli = argparse.ArgumentParser(description='Sequitr worker process')
    cli.add_argument('--job', help='Path to job description file')
    cli.add_argument('--out', help='Path to output folder')
    return 0 if worker(cli.parse_args(argv)) else 1            # the server tells failed jobs from finished ones by this


if __name__ == '__main__':
    sys.exit(main())
