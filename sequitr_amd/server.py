"""Minimal job server: the process the reference's README documents
(README.md:42-81) but does not ship.  Polls JOBDIR every DELAY seconds, runs at most MAX_PROCESSES worker
processes (``python -m sequitr_amd.worker --job X.job --out OUTDIR/JOB_<ID>``, the reference's
``python worker.py --job ... --out ...``, worker.py:307-316), hands each GPU job one of the allowed GPUs
(least loaded first; ``--gpus`` or DEFAULT_GPUS, core.py:41-42) through LOCAL_RANK, which
jobs._resolve_device reads, and renames a finished job file to ``.job.complete`` (worker.py:218-239) -- or to
``.job.failed`` when the worker exits non-zero (its job function raised), ``.job.invalid`` when it cannot be parsed.
``--setup`` writes ``server.config`` (sections config / tensorflow / cpu / gpu, core.py:57-87).

One process per job, one GPU per process: the reference's process model, which is also the MI355X one.
"""
import argparse
import configparser
import logging
import os
import subprocess
import sys
import time

from . import core, serverlogs, worker

logger = logging.getLogger('server_process')


def detect_gpus():
    """Device names without initialising a GPU runtime in the server process (render nodes / KFD topology /
    *_VISIBLE_DEVICES: sequitr_amd/hwinfo.py -- no torch.cuda call, the server forks one worker per job)."""
    from .hwinfo import count_gpus
    return ['/device:GPU:%d' % i for i in range(count_gpus() or 0)]


def setup(config_file='server.config', jobdir='', logdir='', outdir='', modeldir=''):
    """--setup: write a server.config the package's _configure() reads back (core.py:57-87)."""
    gpus = detect_gpus()
    cfg = configparser.ConfigParser()
    cfg.optionxform = str
    cfg['config'] = {'LOGDIR': logdir, 'JOBDIR': jobdir, 'OUTDIR': outdir, 'SERVER_IP': '',
                     'DEFAULT_GPUS': repr(list(range(len(gpus))) or core.ServerConfiguration.DEFAULT_GPUS),
                     'MAX_PROCESSES': str(max(1, len(gpus)) if gpus else core.ServerConfiguration.MAX_PROCESSES),
                     'DELAY': str(core.ServerConfiguration.DELAY), 'LOCAL': 'True', 'VERBOSE_LOG': 'True',
                     'CORES': str(os.cpu_count() or 0)}
    cfg['tensorflow'] = {'TF_LOG_LEVEL': '3', 'LOGDIR': logdir, 'MODELDIR': modeldir,
                         'LOG_DEVICE_PLACEMENT': 'True', 'ALLOW_GROWTH': 'True'}
    cfg['cpu'] = {'cpu0': '/device:CPU:0'}
    cfg['gpu'] = {'gpu%d' % i: g for i, g in enumerate(gpus)}
    with open(config_file, 'w') as f:
        cfg.write(f)
    return config_file


class Server(object):
    def __init__(self, jobdir, outdir, gpus=None, max_processes=None, delay=None, python=None):
        self.jobdir, self.outdir = jobdir, outdir
        self.gpus = list(gpus if gpus is not None else core.ServerConfiguration.DEFAULT_GPUS)
        self.max_processes = int(max_processes or core.ServerConfiguration.MAX_PROCESSES)
        self.delay = core.ServerConfiguration.DELAY if delay is None else delay
        self.python = python or sys.executable
        self.running = {}                                      # job file -> (Popen, gpu or None, JobWrapper)
        self.finished = []                                     # (job ID, return code)
        self.unparsed = {}                                     # job file -> ((size, mtime_ns), polls seen unchanged)
        self.settle = max(float(self.delay), 1.0)              # seconds a broken file must have been left alone

    def _settled(self, fn):
        """True once a file that does not parse has stopped changing: jobs are submitted by dropping files into the
        polled folder, so a file caught mid-write (a non-atomic copy, NFS) must not be discarded -- the rename keeps
        the inode and the finished job would land under .invalid and never run.  Settled = the same (size, mtime) on
        two consecutive polls AND last written at least `settle` seconds ago."""
        try:
            st = os.stat(fn)
        except OSError:
            self.unparsed.pop(fn, None)
            return False
        sig = (st.st_size, st.st_mtime_ns)
        seen, polls = self.unparsed.get(fn, (None, 0))
        polls = polls + 1 if seen == sig else 1
        self.unparsed[fn] = (sig, polls)
        return polls >= 2 and time.time() - st.st_mtime >= self.settle

    def pending(self):
        """settled .job files, highest priority first (job header only: worker.py:42-84)"""
        jobs = []
        for f in sorted(os.listdir(self.jobdir)):
            fn = os.path.join(self.jobdir, f)
            if not f.endswith('.job') or fn in self.running:
                continue
            job = worker.parse_job_file(fn, header_only=True, quiet=fn in self.unparsed)
            if job is not None:
                self.unparsed.pop(fn, None)
                jobs.append(job)
            elif self._settled(fn):                            # logged once here; never polled again
                logger.error('Job file {0} cannot be parsed: renamed to .job.invalid'.format(fn))
                os.rename(fn, fn + '.invalid')
                self.unparsed.pop(fn, None)
        return sorted(jobs, key=lambda j: -int(j.priority))

    def _free_gpu(self):
        load = {g: 0 for g in self.gpus}
        for _, g, _ in self.running.values():
            if g is not None:
                load[g] += 1
        return min(self.gpus, key=lambda g: (load[g], g)) if self.gpus else None

    def reap(self):
        for fn in list(self.running):
            proc, gpu, job = self.running[fn]
            rc = proc.poll()
            if rc is None:
                continue
            del self.running[fn]
            self.finished.append((job.ID, rc))
            logger.info('Job {0} finished with code {1}'.format(job.ID, rc))
            if os.path.exists(fn):
                if rc == 0:
                    job.complete = True                        # X.job -> X.job.complete
                else:                                          # the worker logs and swallows the exception but exits 1
                    os.rename(fn, fn + '.failed')              # X.job -> X.job.failed: visible, not retried for ever

    def launch(self, job):
        gpu = self._free_gpu() if str(job.device).upper() == 'GPU' else None
        out = os.path.join(self.outdir, 'JOB_' + str(job.ID))
        env = dict(os.environ)
        if gpu is not None:
            env['LOCAL_RANK'] = str(gpu)                       # jobs._resolve_device: options['gpu'] or LOCAL_RANK
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        cmd = [self.python, '-m', 'sequitr_amd.worker', '--job', job.filename, '--out', out]
        logger.info('Launching job {0} on {1}: {2}'.format(job.ID, 'GPU %s' % gpu if gpu is not None else 'CPU', ' '.join(cmd)))
        self.running[job.filename] = (subprocess.Popen(cmd, env=env), gpu, job)

    def poll_once(self):
        self.reap()
        for job in self.pending():
            if len(self.running) >= self.max_processes:
                break
            self.launch(job)
        return len(self.running)

    def serve(self, once=False):
        while True:
            busy = self.poll_once()
            if once and not busy and not self.pending() and not self.unparsed:
                return self.finished
            time.sleep(self.delay if not once else min(self.delay, 0.2))


def main(argv=None):
    p = argparse.ArgumentParser(description='Sequitr server process')
    p.add_argument('--jobdir', default=None, help='Path to job directory')
    p.add_argument('--logdir', default=None, help='Path to log directory')
    p.add_argument('--outdir', default=None, help='Path to the job output directory')
    p.add_argument('--gpus', nargs='*', type=int, choices=range(8), default=None,
                   help='Specify the gpus which can be used for processing')
    p.add_argument('--local', action='store_true', help='Running a local server only. Prevents pinging.')
    p.add_argument('--setup', action='store_true', help='Runs the setup configuration to determine hardware specs, '
                   'and generate the config file. On server restart, the config will persist.')
    p.add_argument('--use_config', default='server.config', help='Use a specific (non-default) server config file.')
    p.add_argument('--once', action='store_true', help='Process the jobs present, then exit (tests, batch use).')
    args = p.parse_args(argv)
    if args.setup:
        fn = setup(args.use_config, jobdir=args.jobdir or '', logdir=args.logdir or '', outdir=args.outdir or '')
        print('wrote {0}'.format(fn))
        return 0
    core._configure(args.use_config)
    jobdir = args.jobdir or core.ServerConfiguration.JOBDIR
    outdir = args.outdir or core.ServerConfiguration.OUTDIR or jobdir
    logdir = args.logdir or core.ServerConfiguration.LOGDIR or outdir
    if not jobdir or not os.path.isdir(jobdir):
        raise IOError('Job directory {0!r} does not exist (use --jobdir or server.config)'.format(jobdir))
    serverlogs.setup_logging(logdir, log_name='server_process')
    srv = Server(jobdir, outdir, gpus=args.gpus)
    logger.info('Serving {0} on GPUs {1}, <= {2} workers'.format(jobdir, srv.gpus, srv.max_processes))
    srv.serve(once=args.once)
    serverlogs.shutdown_logging('server_process')
    return 0


if __name__ == '__main__':
    sys.exit(main())
