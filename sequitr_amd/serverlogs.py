"""Logging conventions of sequitr/serverlogs.py: timestamped ``LOG_(date)_time.txt`` in the
job output directory, logger names ``server_process`` / ``worker_process``, and the
``exception_logger`` decorator that logs and SWALLOWS exceptions (serverlogs.py:103-127) --
the error behaviour of the job plugin boundary."""
import functools
import logging
import os
from time import gmtime, strftime

from . import core


_STAMP = "(%Y-%m-%d)_%H-%M-%S"                                 # serverlogs.py:30-32: LOG_(date)_time.txt
_LINE = '[%(levelname)s][%(asctime)s] %(message)s'
_DATE = '%Y/%m/%d %I:%M:%S %p'


def generate_log_filename():
    return "LOG_{0}.txt".format(strftime(_STAMP, gmtime()))


def setup_logging(filepath=None, log_name='server_process'):
    """File + stream handlers on the named logger; returns the log file (serverlogs.py:36-78)."""
    if filepath is None:
        filepath = core.ServerConfiguration.LOGDIR
    if not os.path.exists(filepath):
        raise IOError('LOG_DIR filepath does not exist: {0:s}'.format(filepath))
    if log_name not in core.DEFAULT_LOGGER_PROCESSES:
        raise ValueError('Log_name should be in {0:s}.'.format(str(core.DEFAULT_LOGGER_PROCESSES)))
    target = os.path.join(filepath, generate_log_filename())
    log = logging.getLogger(log_name)
    log.setLevel(logging.DEBUG)
    layout = logging.Formatter(_LINE, datefmt=_DATE)
    for sink in (logging.FileHandler(target), logging.StreamHandler()):
        sink.setFormatter(layout)
        log.addHandler(sink)
    return target


def shutdown_logging(log_name='worker_process'):
    """Detach and close this process's handlers (lets tests and servers reuse the name)."""
    log = logging.getLogger(log_name)
    while log.handlers:
        sink = log.handlers[0]
        log.removeHandler(sink)
        sink.close()


def get_logger(loggers=core.DEFAULT_LOGGER_PROCESSES):
    """ First configured logger of server / worker process, else None (serverlogs.py:81-91) """
    configured = (logging.getLogger(n) for n in loggers)
    return next((lg for lg in configured if lg.handlers), None)


_FAILURES = [0]


def failure_count():
    """number of exceptions exception_logger has swallowed in this process (the worker's exit code reads it)"""
    return _FAILURES[0]


def exception_logger(function):
    """Log any exception of the wrapped call and return None instead of raising (serverlogs.py:113-127); the
    failure is counted so that the worker PROCESS can still report it to the server through its exit code."""
    @functools.wraps(function)
    def guarded(*args, **kwargs):
        try:
            return function(*args, **kwargs)
        except Exception:
            _FAILURES[0] += 1
            where = "There was an exception in: {0:s}".format(function.__name__)
            sink = get_logger()
            (sink.exception if sink is not None else print)(where)
            return None
    return guarded
