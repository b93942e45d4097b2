"""Six-line stand-in for ml_collections.ConfigDict: attribute get/set on a dict."""


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v
