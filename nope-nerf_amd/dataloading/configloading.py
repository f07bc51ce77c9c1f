"""YAML configs layered over a default file (reference dataloading/configloading.py:3-46)."""
import yaml


def _read(path):
    with open(path, 'r') as fh:
        return yaml.load(fh, Loader=yaml.Loader) or {}


def update_recursive(dict1, dict2):
    """Overlay dict2 on dict1 in place; nested dicts merge key by key, everything else is replaced."""
    for key, value in dict2.items():
        if isinstance(value, dict):
            if not isinstance(dict1.get(key), dict):
                dict1[key] = {}
            update_recursive(dict1[key], value)
        else:
            dict1[key] = value


def load_config(path, default_path=None, inherit_from=None):
    """cfg = defaults (from `inherit_from`, itself layered over `default_path`, or from `default_path`) overlaid with `path`."""
    if inherit_from is not None:
        cfg = load_config(inherit_from, default_path)
    elif default_path is not None:
        cfg = _read(default_path)
    else:
        cfg = {}
    update_recursive(cfg, _read(path))
    return cfg
