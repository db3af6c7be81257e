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


class NetConfiguration(object):
    """Generic network configuration (utils.py:247-421)."""

    def __init__(self):
        self.name = 'UNet2D_test'
        self.dropout = 0.5
        self.warm_start = False
        self.shape = (64, 64)
        self.num_inputs = 1
        self.num_outputs = 2
        self.num_epochs = 1000
        self.learning_rate = 0.01
        self.augment = True
        self.path = None
        self.training_data = 'train.tfrecord'
        self.test_data = 'test.tfrecord'
        self.image_dict = {}

    @property
    def name(self):
        return self._name

    @name.setter
    def name(self, name):
        if not isinstance(name, str):
            raise TypeError('Name is not a string.')
        if name not in MODELS:
            raise ValueError('Net name is not recognized.')
        self._name = name

    @property
    def dropout(self):
        return self._dropout

    @dropout.setter
    def dropout(self, dropout):
        if not isinstance(dropout, float):
            raise TypeError('Dropout is not a float.')
        if dropout < 0 or dropout > 1:
            raise ValueError('Dropout should be in the (0-1) range.')
        self._dropout = dropout

    @property
    def shape(self):
        return self._shape

    @shape.setter
    def shape(self, shape):
        if not isinstance(shape, (tuple, list)):
            raise TypeError('Shape is not a tuple.')
        self._shape = tuple(shape)

    @property
    def warm_start(self):
        return self._warm_start

    @warm_start.setter
    def warm_start(self, warm_start):
        if not isinstance(warm_start, bool):
            raise TypeError('Warm start is not a boolean.')
        self._warm_start = warm_start

    @property
    def export_dir_base(self):
        return os.path.join(core.TensorflowConfiguration.MODELDIR, self.name)

    def warm_start_from(self, model_num=None):
        if not self.warm_start:
            return None
        return get_latest_model_dir(self.export_dir_base)

    def get_latest_model_dir(self):
        return get_latest_model_dir(self.export_dir_base)

    @property
    def training_data_file(self):
        if isinstance(self.training_data, list):
            return [os.path.join(self.path, f) for f in self.training_data]
        return os.path.join(self.path, self.training_data)

    @property
    def testing_data_file(self):
        if isinstance(self.test_data, list):
            return [os.path.join(self.path, f) for f in self.test_data]
        return os.path.join(self.path, self.test_data)

    @classmethod
    def from_params(cls, params, preload_model=False):
        """Instantiate from a parameter dict; preload_model first reads the latest saved
        net.config, then the dict overrides it (utils.py:362-388)."""
        if not isinstance(params, dict):
            raise TypeError('Parameters are not specified in dictionary.')
        config = cls()
        if preload_model:
            config.name = params['name']
            config.load()
        for p in params:
            setattr(config, p, params[p])
        return config

    def to_params(self):
        """ {member name without leading underscore: value} (utils.py:390-395) """
        return {m.lstrip('_'): getattr(self, m.lstrip('_')) for m in self.__dict__.keys()}

    def save(self, filename):
        export = {str(self.__class__.__name__): self.to_params()}
        with open(filename, 'w') as f:
            f.write(json.dumps(export, indent=2, separators=(',', ': ')))

    def load(self, filename='net.config'):
        model_dir = self.get_latest_model_dir()
        model_fn = os.path.join(model_dir or '', filename)
        if model_dir is None or not os.path.exists(model_fn):
            raise IOError('Cannot preload config: {0:s}'.format(model_fn))
        with open(model_fn, 'r') as f:
            params = json.load(f)[str(self.__class__.__name__)]
        logger.info('Loading model parameters from: {0:s}'.format(model_fn))
        for p in params:
            setattr(self, p, params[p])


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
