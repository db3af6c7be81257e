"""Job / worker API -- the process boundary the MI355X engine drops in behind.

Mirror of sequitr/worker.py: INI job files with a ``[job]`` section (worker.py:49-60),
``parse_job_file`` -> ``JobWrapper`` (worker.py:42-109), ``JobWrapper.__call__`` importing
``module`` and calling ``func(params, options)`` with ``params['output']`` injected
(worker.py:195-215), completion by renaming ``X.job`` -> ``X.job.complete``
(worker.py:218-239), ``worker(args)`` and the ``--job/--out`` CLI (worker.py:252-316).

New, for the MI355X node: ``device = GPU`` jobs resolve to a HIP device.  The worker's
rank -> GPU mapping is carried in ``options['gpu']`` (set by the server) or LOCAL_RANK;
nothing in the reference read ``device`` after validation.
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


@serverlogs.exception_logger
def parse_job_file(filename, header_only=False):
    """Parse a .job file into a JobWrapper; header_only skips params/options (worker.py:42-109)."""
    if not isinstance(filename, str):
        raise Exception("Job filename is not correctly formed")
    if not filename.endswith('.job'):
        raise IOError('Job {0:s} does not have .job file extenstion'.format(filename))

    cfg = configparser.ConfigParser()
    cfg.optionxform = str                                      # keys are case-sensitive ('ID')
    cfg.read(filename)

    def get(key):
        for k in (key, key.lower()):
            if cfg.has_option('job', k):
                return cfg.get('job', k)
        raise configparser.NoOptionError(key, 'job')

    job = JobWrapper(ID=get('ID'), filename=filename, owner=get('user'),
                     priority=get('priority'), device=get('device'))
    job._module = get('module')
    job._func = get('func')
    if header_only:
        return job

    job._params = literal_eval(get('params'))
    if cfg.has_option('job', 'options'):
        job._options = literal_eval(cfg.get('job', 'options'))
    if cfg.has_option('job', 'lib_path'):
        lib_path = cfg.get('job', 'lib_path')
        if os.path.exists(lib_path):
            job._lib_path = lib_path
    return job


class JobWrapper(object):
    """Specifies, executes, logs and completes one job (worker.py:115-239)."""

    def __init__(self, ID=None, filename=None, owner='root', priority=99, device='CPU'):
        if device not in ['CPU', 'GPU']:
            logging.warning('Device {0:s} not recognised'.format(str(device)))
            raise ValueError
        self._owner = owner
        self._device = device
        self._ID = ID
        self._filename = filename
        self._priority = priority
        self._complete = False
        self._lib_path = None
        self._module = None
        self._func = None
        self._options = {}
        self._params = {}

    @property
    def ID(self): return self._ID
    @property
    def device(self): return self._device
    @property
    def owner(self): return self._owner
    @property
    def filename(self): return self._filename
    @property
    def priority(self): return self._priority

    @property
    def job_output(self):
        return self._job_output

    @job_output.setter
    def job_output(self, job_output):
        if not isinstance(job_output, str):
            raise TypeError('Output folder must be a string')
        check_and_makedir(job_output)
        self._job_output = job_output

    @staticmethod
    def load(filename, header_only=False):
        return parse_job_file(filename, header_only=header_only)

    @serverlogs.exception_logger
    def __call__(self):
        """Import the job module and run func(params, options); the return value is ignored
        and exceptions are logged, not raised (worker.py:195-215, serverlogs.py:113-127)."""
        if self.complete:
            return
        if self._lib_path:
            sys.path.append(self._lib_path)
        m = importlib.import_module(self._module)
        func = getattr(m, self._func)
        self._params['output'] = self.job_output
        self._params.setdefault('device', self._device)       # 'CPU' | 'GPU' for the job function
        func(self._params, self._options)

    @property
    def complete(self): return self._complete

    @complete.setter
    def complete(self, flag=False):
        """JOB_x.job -> JOB_x.job.complete (worker.py:218-239)."""
        if not flag:
            return
        try:
            os.rename(self._filename, self._filename + '.complete')
        except OSError:
            raise OSError('Failed to set job flags to complete')
        self._complete = True


def worker(args, log=True):
    """Run one job: args.job = job file, args.out = output folder (worker.py:252-295)."""
    if not isinstance(args, argparse.Namespace):
        raise TypeError('Args must be of type argsparse.Namespace. The worker'
                        ' function cannot be called directly.')
    if any([a not in args for a in ['out', 'job']]):
        raise AttributeError('Could not find .out or .job in args.')
    check_and_makedir(args.out)
    if log:
        serverlogs.setup_logging(args.out, log_name='worker_process')
        logging.getLogger('worker_process').info(args.job)
    job = JobWrapper.load(args.job)
    if job is not None:
        job.job_output = args.out
        job()
    if log:
        serverlogs.shutdown_logging('worker_process')
    return


def main(argv=None):
    parser = argparse.ArgumentParser(description='Sequitr worker process')
    parser.add_argument('--job', help='Path to job description file')
    parser.add_argument('--out', help='Path to output folder')
    worker(parser.parse_args(argv))


if __name__ == '__main__':
    main()
