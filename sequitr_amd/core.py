"""Server / back-end configuration -- mirror of sequitr/core.py (INI `server.config`,
sections config / tensorflow / cpu / gpu; core.py:34-106).

Differences, all documented in DESIGN.md: Python 3 ``configparser``; a missing
``server.config`` is not an error (the reference logs through an undefined logger,
core.py:60-63); ``DEFAULT_GPUS`` defaults to the 8 GPUs of one MI355X node.
The ``tensorflow`` section keeps its name for file compatibility; it now configures the
HIP back end's LOGDIR / MODELDIR.
"""
import configparser
import os
from ast import literal_eval

__version__ = '0.1.7'                                          # reference version tracked

DEFAULT_LOGGER_PROCESSES = ('server_process', 'worker_process')


class ServerConfiguration:
    """ A server config (core.py:34-46) """
    VERSION = __version__
    LOGDIR = ''
    JOBDIR = ''
    OUTDIR = ''
    SERVER_IP = ''
    DEFAULT_GPUS = [0, 1, 2, 3, 4, 5, 6, 7]
    MAX_PROCESSES = 8
    DELAY = 60
    LOCAL = True
    VERBOSE_LOG = True
    CORES = 0
    CPUS = []
    GPUS = []


class TensorflowConfiguration:
    """ Back-end config; name kept for job/config-file compatibility (core.py:48-54) """
    TF_LOG_LEVEL = '3'
    LOGDIR = ''
    MODELDIR = ''
    LOG_DEVICE_PLACEMENT = True
    ALLOW_GROWTH = True


BOOL_OPTS = ('LOCAL', 'VERBOSE_LOG', 'LOG_DEVICE_PLACEMENT', 'ALLOW_GROWTH')
INT_OPTS = ('MAX_PROCESSES', 'DELAY', 'CORES')
LIST_OPTS = ('DEFAULT_GPUS',)


def _get_config_opt_correct_type(config, section, opt):
    """ Return correctly typed configuration info (core.py:90-106) """
    if opt.upper() in BOOL_OPTS:
        return config.getboolean(section, opt)
    if opt.upper() in INT_OPTS:
        return config.getint(section, opt)
    if opt.upper() in LIST_OPTS:
        return literal_eval(config.get(section, opt))
    return config.get(section, opt)


def _configure(config_file='server.config'):
    """ Configure the package from a config file; returns the version, or None if absent """
    if not os.path.exists(config_file):
        return None
    config = configparser.ConfigParser()
    config.read(config_file)
    for section, target in (('config', ServerConfiguration), ('tensorflow', TensorflowConfiguration)):
        if config.has_section(section):
            for opt in config.options(section):
                setattr(target, opt.upper(), _get_config_opt_correct_type(config, section, opt))
    ServerConfiguration.CPUS = [config.get('cpu', c) for c in config.options('cpu')] \
        if config.has_section('cpu') else []
    ServerConfiguration.GPUS = [config.get('gpu', g) for g in config.options('gpu')] \
        if config.has_section('gpu') else []
    return ServerConfiguration.VERSION


_configure()
