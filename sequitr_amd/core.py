"""Server / back-end configuration -- mirror of sequitr/core.py (INI `server.config`,
sections config / tensorflow / cpu / gpu; core.py:34-106).

Differences, all documented in HISTORY.md (section 8): Python 3 ``configparser``; a missing
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


def _settings(name, doc, **defaults):
    """A namespace class of upper-case settings: read as ``Name.SETTING`` everywhere, overwritten by _configure()."""
    return type(name, (object,), dict(defaults, __doc__=doc))


# the settings and their defaults (core.py:34-54); DEFAULT_GPUS / MAX_PROCESSES sized for one 8-GPU MI355X node
ServerConfiguration = _settings(
    'ServerConfiguration', 'server side: folders, address, worker slots (core.py:34-46)',
    VERSION=__version__, LOGDIR='', JOBDIR='', OUTDIR='', SERVER_IP='', DEFAULT_GPUS=list(range(8)), MAX_PROCESSES=8,
    DELAY=60, LOCAL=True, VERBOSE_LOG=True, CORES=0, CPUS=[], GPUS=[])
TensorflowConfiguration = _settings(
    'TensorflowConfiguration', 'back end; the name is kept for job / config-file compatibility (core.py:48-54)',
    TF_LOG_LEVEL='3', LOGDIR='', MODELDIR='', LOG_DEVICE_PLACEMENT=True, ALLOW_GROWTH=True)

_SECTIONS = {'config': ServerConfiguration, 'tensorflow': TensorflowConfiguration}


def _typed(parser, section, option, default):
    """The option parsed as the type of the setting's default (the reference keeps name lists per type,
    core.py:90-106: same result for every setting it lists); unknown settings stay strings."""
    if isinstance(default, bool):
        return parser.getboolean(section, option)
    if isinstance(default, int):
        return parser.getint(section, option)
    if isinstance(default, (list, tuple)):
        return literal_eval(parser.get(section, option))
    return parser.get(section, option)


def _configure(config_file='server.config'):
    """Apply an INI file to the settings above (core.py:57-87); returns the version, or None when there is no file."""
    if not os.path.exists(config_file):
        return None
    parser = configparser.ConfigParser()
    parser.read(config_file)
    for section, target in _SECTIONS.items():
        for option in (parser.options(section) if parser.has_section(section) else ()):
            key = option.upper()
            setattr(target, key, _typed(parser, section, option, getattr(target, key, '')))
    for section, key in (('cpu', 'CPUS'), ('gpu', 'GPUS')):     # device lists: one entry per line of the section
        names = parser.options(section) if parser.has_section(section) else ()
        setattr(ServerConfiguration, key, [parser.get(section, n) for n in names])
    return ServerConfiguration.VERSION


_configure()
