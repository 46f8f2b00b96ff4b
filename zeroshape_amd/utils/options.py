"""Attribute-and-item dict used for ``opt`` / ``var`` (same behaviour as the reference's
utils/util.py:378-412 EasyDict: nested dicts become EasyDicts, attributes and items are
the same storage).  Only what the hot-path functions need; the reference's CLI / yaml
option system (utils/options.py) is out of scope."""


class EasyDict(dict):
    def __init__(self, d=None, **kwargs):
        super().__init__()
        d = dict(d or {})
        d.update(kwargs)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, name, value):
        if isinstance(value, (list, tuple)):
            value = [self.__class__(x) if isinstance(x, dict) else x for x in value]
        elif isinstance(value, dict) and not isinstance(value, self.__class__):
            value = self.__class__(value)
        super().__setattr__(name, value)
        super().__setitem__(name, value)

    __setitem__ = __setattr__

    def update(self, e=None, **f):
        d = dict(e or {})
        d.update(f)
        for k in d:
            setattr(self, k, d[k])

    def pop(self, k, d=None):
        if hasattr(self, k):
            delattr(self, k)
        return super().pop(k, d)
