"""Minimal hyper-parameter container with the interface the reference drivers use.

Mirrors the subset of utils/hparams.py (a port of tf.contrib.training.HParams) that train.py /
eval.py / synthesize.py touch: attribute access, parse("a=1,b=2.5,c=true"), values(), to_json(),
override_from_dict(), set_hparam(), add_hparam(), get(), `in`.  Scalars and flat lists only.
"""
import json
import re


def _cast(name, old, value):
    """Cast `value` (str or python scalar) to the type of the existing value `old`."""
    if isinstance(old, bool):
        if isinstance(value, str):
            v = value.strip().lower()
            if v in ("true", "1"):
                return True
            if v in ("false", "0"):
                return False
            raise ValueError("Could not parse hparam '%s' bool value '%s'" % (name, value))
        return bool(value)
    if isinstance(old, int):
        if isinstance(value, str):
            return int(value)
        if isinstance(value, float) and value != int(value):
            raise ValueError("Could not cast hparam '%s' of type int from %r" % (name, value))
        return int(value)
    if isinstance(old, float):
        return float(value)
    if isinstance(old, str):
        return str(value)
    return value


class HParams(object):
    def __init__(self, **kwargs):
        object.__setattr__(self, "_types", {})
        for k, v in kwargs.items():
            self.add_hparam(k, v)

    def add_hparam(self, name, value):
        if getattr(self, name, None) is not None and name in self._types:
            raise ValueError("Hyperparameter name is reserved: %s" % name)
        self._types[name] = (type(value[0]) if isinstance(value, (list, tuple)) and value else type(value),
                             isinstance(value, (list, tuple)))
        object.__setattr__(self, name, list(value) if isinstance(value, tuple) else value)

    def set_hparam(self, name, value):
        if name not in self._types:
            raise ValueError("Unknown hyperparameter: %s" % name)
        old = getattr(self, name)
        if isinstance(old, list):
            if not isinstance(value, list):
                raise ValueError("Must pass a list for multi-valued parameter: %s." % name)
            proto = old[0] if old else ""
            object.__setattr__(self, name, [_cast(name, proto, v) for v in value])
        else:
            if isinstance(value, list):
                raise ValueError("Must not pass a list for single-valued parameter: %s" % name)
            object.__setattr__(self, name, _cast(name, old, value))

    def override_from_dict(self, values_dict):
        for k, v in values_dict.items():
            self.set_hparam(k, v)
        return self

    def parse(self, values):
        """'name=value,name2=[a,b]' -> overrides (same grammar subset as the reference)."""
        pos = 0
        pattern = re.compile(r"\s*(?P<name>[a-zA-Z_]\w*)\s*=\s*(?:\[(?P<list>[^\]]*)\]|(?P<val>[^,\[]*))\s*(?:,|$)")
        out = {}
        while pos < len(values):
            m = pattern.match(values, pos)
            if not m:
                raise ValueError("Malformed hyperparameter value: %s" % values[pos:])
            pos = m.end()
            name = m.group("name")
            if name not in self._types:
                raise ValueError("Unknown hyperparameter type for %s" % name)
            if m.group("list") is not None:
                out[name] = [s.strip() for s in m.group("list").split(",") if s.strip()]
            else:
                out[name] = m.group("val").strip()
        return self.override_from_dict(out)

    def values(self):
        return {k: getattr(self, k) for k in self._types}

    def get(self, key, default=None):
        return getattr(self, key) if key in self._types else default

    def __contains__(self, key):
        return key in self._types

    def to_json(self, indent=None, separators=None, sort_keys=False):
        return json.dumps(self.values(), indent=indent, separators=separators, sort_keys=sort_keys)

    def __str__(self):
        return str(sorted(self.values().items()))
