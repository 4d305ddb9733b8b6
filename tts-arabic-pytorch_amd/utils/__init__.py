"""Config helpers with the reference's names (utils/__init__.py:9-40): `DictConfig`,
`get_custom_config`, `get_basic_config`, `get_config`, `read_lines_from_file`."""
import os

import yaml

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class DictConfig(object):
    def __init__(self, config_dict):
        self.__dict__.update(config_dict)

    def __str__(self):
        return '\n'.join(f"{key}: {val}" for key, val in self.__dict__.items())

    __repr__ = __str__


def get_custom_config(fname):
    with open(fname, 'r') as stream:
        return DictConfig(yaml.safe_load(stream))


def get_basic_config():
    # the reference opens the cwd-relative 'configs/basic.yaml' (utils/__init__.py:31-32);
    # fall back to the copy shipped with this package when run from elsewhere
    path = 'configs/basic.yaml'
    if not os.path.exists(path):
        path = os.path.join(_PKG_ROOT, 'configs', 'basic.yaml')
    return get_custom_config(path)


def get_config(fname):
    config = get_basic_config()
    config.__dict__.update(get_custom_config(fname).__dict__)
    return config


def read_lines_from_file(path, encoding='utf-8'):
    with open(path, 'r', encoding=encoding) as f:
        return [line.strip() for line in f]


def write_lines_to_file(path, lines, mode='w', encoding='utf-8'):
    with open(path, mode, encoding=encoding) as f:
        for i, line in enumerate(lines):
            f.write(line if i == len(lines) - 1 else line + '\n')
