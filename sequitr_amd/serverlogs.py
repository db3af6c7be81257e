"""Logging conventions of sequitr/serverlogs.py: timestamped ``LOG_(date)_time.txt`` in the
job output directory, logger names ``server_process`` / ``worker_process``, and the
``exception_logger`` decorator that logs and SWALLOWS exceptions (serverlogs.py:103-127) --
the error behaviour of the job plugin boundary."""
import functools
import logging
import os
from time import gmtime, strftime

from . import core


def generate_log_filename():
    """ Return a timestamped log filename (serverlogs.py:30-32) """
    return "LOG_" + strftime("(%Y-%m-%d)_%H-%M-%S", gmtime()) + ".txt"


def setup_logging(filepath=None, log_name='server_process'):
    """File + stream handlers on the named logger; returns the log file (serverlogs.py:36-78)."""
    if filepath is None:
        filepath = core.ServerConfiguration.LOGDIR
    if not os.path.exists(filepath):
        raise IOError('LOG_DIR filepath does not exist: {0:s}'.format(filepath))
    if log_name not in core.DEFAULT_LOGGER_PROCESSES:
        raise ValueError('Log_name should be in {0:s}.'.format(str(core.DEFAULT_LOGGER_PROCESSES)))
    log_file = os.path.join(filepath, generate_log_filename())
    fmt = logging.Formatter('[%(levelname)s][%(asctime)s] %(message)s', datefmt='%Y/%m/%d %I:%M:%S %p')
    logger = logging.getLogger(log_name)
    for handler in (logging.FileHandler(log_file), logging.StreamHandler()):
        handler.setFormatter(fmt)
        logger.addHandler(handler)
    logger.setLevel(logging.DEBUG)
    return log_file


def shutdown_logging(log_name='worker_process'):
    """Detach and close this process's handlers (lets tests and servers reuse the name)."""
    logger = logging.getLogger(log_name)
    for h in list(logger.handlers):
        logger.removeHandler(h)
        h.close()


def get_logger(loggers=core.DEFAULT_LOGGER_PROCESSES):
    """ First configured logger of server / worker process, else None (serverlogs.py:81-91) """
    for name in loggers:
        logger = logging.getLogger(name)
        if logger.handlers:
            return logger
    return None


def exception_logger(function):
    """Log any exception of the wrapped call and return None instead of raising."""
    @functools.wraps(function)
    def wrapper(*args, **kwargs):
        logger = get_logger()
        try:
            return function(*args, **kwargs)
        except Exception:
            err = "There was an exception in: {0:s}".format(function.__name__)
            if logger is not None:
                logger.exception(err)
            else:
                print(err)
    return wrapper
