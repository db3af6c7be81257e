"""Model/config surface -- mirror of the parts of sequitr/utils.py the hot path touches:
``NetConfiguration`` (utils.py:247-421: fields, ``from_params`` / ``to_params`` /
``save`` / ``load`` with the ``net.config`` JSON keyed by class name), the numbered model
directory helpers (utils.py:143-181, 336-348) and small numeric helpers
(utils.py:105-129, 230-240).

The reference's defects are NOT reproduced: the name whitelist is read from
``./models.txt`` only when that file exists (utils.py:584-593 crashes without it) and the
defaults are always accepted; ``dropout`` stores and returns its value (utils.py:307-315
returns the name).  TensorFlow checkpoints are replaced by ``weights.npz`` holding the
state dict under the reference's variable-scope names.
"""
import json
import logging
import os

import numpy as np

from . import core

logger = logging.getLogger('worker_process')

DEFAULT_MODELS = ('UNet2D_test', 'UNet2d_test', 'GAN_competition', 'competition_GAN')


def get_model_list(path='./models.txt'):
    """White-listed model names: defaults + the lines of ./models.txt if present."""
    models = list(DEFAULT_MODELS)
    if os.path.exists(path):
        with open(path, 'r') as f:
            models += [m.strip() for m in f.readlines() if m.strip()]
    return tuple(models)


MODELS = get_model_list()


def filter_doubling(start_filters=8, num_layers=7, max_filters=4096, reverse=False):
    """[start * 2^i capped at max] for i < num_layers (utils.py:105-123 / networks.common)."""
    f = [min(start_filters * (2 ** i), max_filters) for i in range(num_layers)]
    if reverse:
        f.reverse()
    return f


def check_and_makedir(folder_name):
    """ Does a directory exist? if not create it (returns whether it existed). """
    if not os.path.isdir(folder_name):
        logger.info('Creating output folder {0:s}...'.format(folder_name))
        os.mkdir(folder_name)
        return False
    return True


def _export_dir_fn(x):
    return '{0:d}'.format(x).zfill(4)


def get_latest_model_dir(export_dir_base):
    """ Highest-numbered sub-folder, or None (utils.py:143-156) """
    if not os.path.isdir(export_dir_base):
        return None
    models = [f for f in os.listdir(export_dir_base)
              if os.path.isdir(os.path.join(export_dir_base, f)) and f.isdigit()]
    if not models:
        return None
    return os.path.join(export_dir_base, _export_dir_fn(max(int(f) for f in models)))


def create_new_export_dir(export_dir_base):
    """ Next numbered model export directory: 0001, 0002, ... (utils.py:159-181) """
    check_and_makedir(export_dir_base)
    latest = get_latest_model_dir(export_dir_base)
    num = 0 if latest is None else int(os.path.split(latest)[1])
    new_dir = os.path.join(export_dir_base, _export_dir_fn(num + 1))
    if check_and_makedir(new_dir):
        raise IOError('New export model dir already exists?!?!')
    return new_dir


def power_of_two(number):
    return number > 0 and (number & (number - 1)) == 0


def divisible_by_two_n_times(x, n):
    """ Can x be halved n times (n max-pool levels)?  utils.py:234-240 """
    for _ in range(n):
        x = x / 2.0
    return x % 1 == 0


# the validated fields of the configuration surface (utils.py:281-330): accepted type(s), what a value must satisfy, the
# error raised for each, and how the value is stored.  Everything else is a plain attribute (utils.py:262-278).
_RULES = {
    'name': (str, 'Name is not a string.', lambda v: v in MODELS, 'Net name is not recognized.', None),
    'dropout': (float, 'Dropout is not a float.', lambda v: 0 <= v <= 1, 'Dropout should be in the (0-1) range.', None),
    'shape': ((tuple, list), 'Shape is not a tuple.', None, None, tuple),
    'warm_start': (bool, 'Warm start is not a boolean.', None, None, None),
}
# insertion order = the key order of net.config (the reference writes its members in assignment order)
_DEFAULTS = (('name', 'UNet2D_test'), ('dropout', 0.5), ('warm_start', False), ('shape', (64, 64)), ('num_inputs', 1),
             ('num_outputs', 2), ('num_epochs', 1000), ('learning_rate', 0.01), ('augment', True), ('path', None),
             ('training_data', 'train.tfrecord'), ('test_data', 'test.tfrecord'), ('image_dict', None))


class NetConfiguration(object):
    """Generic network configuration (utils.py:247-421): a bag of named settings, four of them validated on
    assignment, written to / read from ``net.config`` as ``{class name: {setting: value}}``.  Schema-driven here
    (_RULES / _DEFAULTS) instead of one property pair per field; the surface -- attribute names, accepted values,
    exception types and texts, JSON layout -- is the reference's."""

    def __init__(self):
        for key, value in _DEFAULTS:
            setattr(self, key, {} if key == 'image_dict' else value)

    def __setattr__(self, key, value):
        rule = _RULES.get(key)
        if rule is not None:
            kinds, not_kind, accept, not_accepted, store = rule
            if not isinstance(value, kinds) or (kinds is float and isinstance(value, bool)):
                raise TypeError(not_kind)
            if accept is not None and not accept(value):
                raise ValueError(not_accepted)
            if store is not None:
                value = store(value)
        object.__setattr__(self, key, value)

    # -- locations ---------------------------------------------------------------------------------
    @property
    def export_dir_base(self):
        return os.path.join(core.TensorflowConfiguration.MODELDIR, self.name)

    def get_latest_model_dir(self):
        return get_latest_model_dir(self.export_dir_base)

    def warm_start_from(self, model_num=None):
        return self.get_latest_model_dir() if self.warm_start else None

    def _in_path(self, names):
        if isinstance(names, list):
            return [os.path.join(self.path, n) for n in names]
        return os.path.join(self.path, names)

    training_data_file = property(lambda self: self._in_path(self.training_data))
    testing_data_file = property(lambda self: self._in_path(self.test_data))

    # -- dict / file round trip (utils.py:362-421) ---------------------------------------------------
    def update(self, params):
        for key, value in params.items():
            setattr(self, key, value)
        return self

    @classmethod
    def from_params(cls, params, preload_model=False):
        """preload_model: the latest saved net.config of params['name'] is read first, then params override it."""
        if not isinstance(params, dict):
            raise TypeError('Parameters are not specified in dictionary.')
        config = cls()
        if preload_model:
            config.name = params['name']
            config.load()
        return config.update(params)

    def to_params(self):
        return dict(vars(self))

    def save(self, filename):
        with open(filename, 'w') as f:
            json.dump({type(self).__name__: self.to_params()}, f, indent=2, separators=(',', ': '))

    def load(self, filename='net.config'):
        model_dir = self.get_latest_model_dir()
        source = os.path.join(model_dir or '', filename)
        if model_dir is None or not os.path.exists(source):
            raise IOError('Cannot preload config: {0:s}'.format(source))
        with open(source, 'r') as f:
            stored = json.load(f)[type(self).__name__]
        logger.info('Loading model parameters from: {0:s}'.format(source))
        self.update(stored)


WEIGHTS_FILE = 'weights.npz'


def save_model(state_dict, config):
    """Write a new numbered model dir with weights.npz + net.config; stands in for
    save_estimator_model's checkpoint copy (utils.py:186-223).  Returns the directory."""
    if not isinstance(config, NetConfiguration):
        raise TypeError('Configurations needs to be of type NetConfiguration')
    export_dir = create_new_export_dir(config.export_dir_base)
    logger.info('Saving model: {0:s}'.format(export_dir))
    np.savez(os.path.join(export_dir, WEIGHTS_FILE), **{k: np.asarray(v) for k, v in state_dict.items()})
    config.save(os.path.join(export_dir, 'net.config'))
    return export_dir


def load_model_weights(model_dir):
    """{variable name: ndarray} from a numbered model dir (numpy.load, no pickle)."""
    with np.load(os.path.join(model_dir, WEIGHTS_FILE), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def __getattr__(name):
    """``utils.CentroidWriter`` (sequitr/utils.py:479-578) lives in sequitr_amd.centroids: it needs the HIP
    library, which this module must not import just to read a config."""
    if name == 'CentroidWriter':
        from .centroids import CentroidWriter
        return CentroidWriter
    raise AttributeError("module {0!r} has no attribute {1!r}".format(__name__, name))
