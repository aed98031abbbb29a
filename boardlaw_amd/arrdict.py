"""Dict-of-tensors containers with the surface boardlaw's worlds, agents and search rely on.

Same contract as the reference's rebar.dotdict / rebar.arrdict (rebar/dotdict.py:17-29, rebar/arrdict.py:33-48,
123-148): attribute access to keys, attribute *delegation* to the leaves (`d.cuda()`, `d.clone()`, `d.shape`),
tensor-style indexing that fans out over the leaves, assignment of one arrdict into a slice of another, elementwise
binary operators, `stack`/`cat`, and `namedarrtuple` for fixed-field containers such as the Hex world.
Written from that contract; the implementation is this project's own."""
import operator
from collections import OrderedDict

import numpy as np

try:
    import torch
except ModuleNotFoundError:  # pragma: no cover
    torch = None


def _is_field(key):
    return isinstance(key, str) or (isinstance(key, tuple) and len(key) > 0 and all(isinstance(k, str) for k in key))


class dotdict(OrderedDict):
    """Ordered dict whose keys are also attributes; unknown attributes are looked up on every value."""

    def __getattr__(self, name):
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        if name in self:
            return self[name]
        try:
            return type(self)((k, getattr(v, name)) for k, v in self.items())
        except AttributeError:
            raise AttributeError(f"There is no member called '{name}' and one of the leaves has no attribute '{name}'") from None

    def __call__(self, *args, **kwargs):
        return type(self)((k, v(*args, **kwargs)) for k, v in self.items())

    def __dir__(self):
        return sorted(set(list(super().__dir__()) + [k for k in self.keys() if isinstance(k, str)]))

    def __repr__(self):
        return _describe(self)

    __str__ = __repr__

    def __getstate__(self):
        return self

    def __setstate__(self, state):
        self.update(state)

    def copy(self):
        return type(self)(**self)

    def pipe(self, f, *args, **kwargs):
        return f(self, *args, **kwargs)

    def map(self, f, *args, **kwargs):
        return mapping(f)(self, *args, **kwargs)

    def starmap(self, f, *args, **kwargs):
        return starmapping(f)(self, *args, **kwargs)


def _describe(d, indent=0):
    pad = ' ' * indent
    lines = [f'{type(d).__name__}:']
    width = max([len(str(k)) for k in d.keys()] + [0]) + 4
    for k, v in d.items():
        if isinstance(v, dotdict):
            body = _describe(v, indent + width).splitlines()
            lines.append(f'{str(k):<{width}}{body[0]}')
            lines.extend(' ' * width + l for l in body[1:])
        elif hasattr(v, 'shape') and hasattr(v, 'dtype'):
            lines.append(f'{str(k):<{width}}{type(v).__name__}({tuple(v.shape)}, {v.dtype})')
        else:
            text = str(v).splitlines() or ['']
            lines.append(f'{str(k):<{width}}{text[0][:100]}')
    return ('\n' + pad).join(lines)


def mapping(f):
    """Lifts f to act on the leaves of (nested) dicts; a string names a method of the leaves."""
    def lifted(x, *args, **kwargs):
        if isinstance(x, dict):
            return type(x)((k, lifted(v, *args, **kwargs)) for k, v in x.items())
        if isinstance(f, str):
            return getattr(x, f)(*args, **kwargs)
        return f(x, *args, **kwargs)
    lifted.__name__ = getattr(f, '__name__', str(f))
    lifted.__doc__ = getattr(f, '__doc__', None)
    return lifted


def starmapping(f):
    """Like mapping, but walks several dicts with identical keys in step."""
    def lifted(x, *others):
        if isinstance(x, dict):
            return type(x)((k, lifted(x[k], *(o[k] for o in others))) for k in x)
        if isinstance(f, str):
            return getattr(x, f)(*others)
        return f(x, *others)
    lifted.__name__ = getattr(f, '__name__', str(f))
    return lifted


def leaves(t):
    if isinstance(t, dict):
        return [l for v in t.values() for l in leaves(v)]
    return [t]


def first_value(t):
    while isinstance(t, dict):
        t = next(iter(t.values()))
    return t


class arrdict(dotdict):
    """dotdict whose values are arrays/tensors (or nested arrdicts): indexing and arithmetic fan out to the leaves."""

    def __getitem__(self, key):
        if isinstance(key, str):
            return super().__getitem__(key)
        return type(self)((k, v[key]) for k, v in self.items())

    def __setitem__(self, key, value):
        if _is_field(key):
            super().__setitem__(key, value)
        elif isinstance(value, type(self)) or isinstance(value, arrdict):
            for k in self:
                self[k][key] = value[k]
        else:
            raise ValueError('Setting items must be done with a string key or by passing an arrdict')

    def _binary(self, name, rhs):
        if isinstance(rhs, dict):
            return type(self)((k, getattr(v, name)(rhs[k])) for k, v in self.items())
        return type(self)((k, getattr(v, name)(rhs)) for k, v in self.items())


def _install_operators():
    names = ['lt', 'le', 'eq', 'ne', 'ge', 'gt', 'add', 'sub', 'mul', 'matmul', 'truediv', 'floordiv', 'mod', 'pow',
             'lshift', 'rshift', 'and', 'or', 'xor', 'radd', 'rsub', 'rmul', 'rmatmul', 'rtruediv', 'rfloordiv', 'rmod',
             'rpow', 'rand', 'ror', 'rxor']
    for nm in names:
        dunder = f'__{nm}__'

        def op(self, rhs, _d=dunder):
            return self._binary(_d, rhs)
        op.__name__ = dunder
        setattr(arrdict, dunder, op)
    arrdict.__hash__ = None


_install_operators()


def namedarrtuple(name='AnonymousNamedArrTuple', fields=()):
    """An arrdict subclass with a fixed set of fields (the reference's worlds are built on this)."""
    fields = tuple(fields)

    def __init__(self, *args, **kwargs):
        arrdict.__init__(self, *args, **kwargs)
        if set(fields) != set(self.keys()):
            raise KeyError(f'This NamedArrTuple subclass must be created with exactly the fields {fields}')

    def __setitem__(self, key, value):
        if _is_field(key) and key not in fields:
            raise KeyError(f'Key "{key}" is not in this immutable NamedArrTuple, and so cannot be added')
        arrdict.__setitem__(self, key, value)

    def __delitem__(self, key):
        raise KeyError('Cannot delete keys from this immutable NameArrTuple subclass')

    return type(name, (arrdict,), {'__init__': __init__, '__setitem__': __setitem__, '__delitem__': __delitem__})


def _combine(xs, torch_f, np_f, args, kwargs):
    head = xs[0]
    if isinstance(head, dict):
        return type(head)((k, _combine([x[k] for x in xs], torch_f, np_f, args, kwargs)) for k in head.keys())
    if torch is not None and isinstance(head, torch.Tensor):
        return torch_f(list(xs), *args, **kwargs)
    if isinstance(head, np.ndarray):
        return np_f(list(xs), *args, **kwargs)
    if np.isscalar(head):
        return np.array(xs)
    raise ValueError(f"Can't combine {type(head)}")


def stack(xs, *args, **kwargs):
    return _combine(xs, torch.stack if torch else None, np.stack, args, kwargs)


def cat(xs, *args, **kwargs):
    return _combine(xs, torch.cat if torch else None, np.concatenate, args, kwargs)


@mapping
def clone(t):
    if hasattr(t, 'clone'):
        return t.clone()
    if hasattr(t, 'copy'):
        return t.copy()
    return t


@mapping
def torchify(a):
    if hasattr(a, 'torchify'):
        return a.torchify()
    a = np.asarray(a)
    if np.issubdtype(a.dtype, np.floating):
        dtype = torch.float
    elif np.issubdtype(a.dtype, np.integer):
        dtype = torch.int
    elif np.issubdtype(a.dtype, np.bool_):
        dtype = torch.bool
    else:
        raise ValueError(f"Can't handle {type(a)}")
    return torch.as_tensor(np.array(a), dtype=dtype)


@mapping
def numpyify(t):
    if isinstance(t, tuple):
        return tuple(numpyify(x) for x in t)
    if torch is not None and isinstance(t, torch.Tensor):
        return t.clone().detach().cpu().numpy()
    if hasattr(t, 'numpyify'):
        return t.numpyify()
    return t


def from_dicts(t):
    if isinstance(t, dict):
        return arrdict((k, from_dicts(v)) for k, v in t.items())
    return t


def to_dicts(t):
    if isinstance(t, dict):
        return {k: to_dicts(v) for k, v in t.items()}
    return t
