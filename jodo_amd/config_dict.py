"""Attribute-style config container used when `ml_collections` is not installed.

The reference's config files (configs/vpsde_*.py) only use attribute get/set on
`ml_collections.ConfigDict`; this class provides the same surface so our configs/ mirror the
reference key-for-key and a reference config object can be passed to our modules unchanged.
"""


class ConfigDict(dict):
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as exc:
            raise AttributeError(key) from exc

    def __setattr__(self, key, value):
        self[key] = value

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, ConfigDict) else v) for k, v in self.items()}


def get_config_dict_class():
    try:
        import ml_collections
        return ml_collections.ConfigDict
    except Exception:
        return ConfigDict
