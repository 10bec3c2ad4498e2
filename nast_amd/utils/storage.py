"""Attribute dictionary with the reference's `Storage` behaviour on the keys the hot path uses
(utils/storage.py:3-84): attribute access, missing attribute -> AttributeError, `a - b` key difference."""


class Storage(dict):
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError("This storage has no attribute %s" % key)

    def __setattr__(self, key, value):
        self[key] = value

    def __delattr__(self, key):
        del self[key]

    def __sub__(self, other):
        res = Storage()
        for k, v in self.items():
            if k not in other:
                res[k] = v
            elif isinstance(v, Storage) and isinstance(other[k], Storage):
                res[k] = v - other[k]
        return res

    def update(self, other):
        """Like the reference: does NOT overwrite existing keys."""
        for k, v in other.items():
            if k not in self:
                self[k] = v
            elif isinstance(self[k], Storage) and isinstance(v, Storage):
                self[k].update(v)
