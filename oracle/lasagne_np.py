"""NumPy stand-in for the part of the Lasagne API the reference's ``build_ca`` functions call
(test infrastructure; see ``oracle/__init__.py``).

Why this exists.  The reference's networks are built by calls into Lasagne (git master, unpinned) and
evaluated by Theano 0.9.0 (``requirements.txt:1-2``); neither is vendored, installed or installable
here, so the network arithmetic cannot be taken from the reference itself.  What CAN be taken from
the reference is the *graph construction*: ``oracle/ref_exec.py`` executes the reference's own
``build_ca`` source (and its mask expressions) with this module standing in for ``lasagne``.  Which
layers exist, how they are wired (the ``l_fc12`` alias of ``separate_dsd.py:228``), every filter size
and stride, and the parameter order ``set_all_param_values`` expects then come from the reference's
code, not from a reading of it; only the semantics of the nine layer classes below are restated
third-party behaviour:

  InputLayer, Conv2DLayer, BiasLayer, MaxPool2DLayer, DenseLayer, ReshapeLayer, InverseLayer,
  ConcatLayer, NonlinearityLayer; helper functions get_all_layers / get_all_params /
  get_all_param_values / set_all_param_values / get_output; nonlinearities.rectify

Restated from the published Lasagne sources (``lasagne/layers/{base,conv,dense,pool,shape,special,
merge,helper}.py``) and Theano's documented op semantics:

* ``Conv2DLayer``: ``W (num_filters, in_channels, fh, fw)``, ``b (num_filters,)``; ``pad='valid'``;
  ``flip_filters=True`` -> ``theano.tensor.nnet.conv2d(filter_flip=True)``, a true convolution:
  ``out[b,o,y,x] = sum_{c,u,v} W[o,c,u,v] * in[b,c, y*sy + fh-1-u, x*sx + fw-1-v] + b[o]``.
  Its default nonlinearity is ``rectify``; the reference passes ``nonlinearity=None`` (identity).
* ``BiasLayer``: ``b (C,)`` added along axis 1 (``shared_axes='auto'``).
* ``MaxPool2DLayer(pool_size)``: stride = pool size, ``ignore_border=True``.
* ``DenseLayer(num_units)``: flattens trailing axes, ``W (num_inputs, num_units)``, ``b (num_units,)``,
  default nonlinearity ``rectify``.
* ``InverseLayer(incoming, layer)``: a ``MergeLayer`` over ``[incoming, layer, layer.input_layer]`` whose
  output is ``theano.grad(None, wrt=layer_in, known_grads={layer_out: incoming})`` -- the
  vector-Jacobian product of ``layer`` at its forward input.  For the max-pool, Theano 0.9's CPU
  ``MaxPoolGrad`` adds the incoming value to EVERY position equal to the window maximum
  (``tie_mode='all'``); cuDNN routes it to the first (``'first'``).
* ``get_all_layers``: depth-first, a layer is emitted after all its incoming layers, each once;
  ``get_all_params``: the layers' parameters in that order, duplicates removed; within a layer ``W``
  before ``b`` (the order of ``add_param`` calls).

Everything is evaluated eagerly in float64 (Theano's CPU default ``floatX``): an ``InputLayer`` holds the
ndarray passed as ``input_var`` and ``get_output`` walks the graph.  Convolution and its VJP are written as
per-tap gather / scatter sums over strided views -- a formulation independent of ``oracle.net_ref``
(torch ``conv2d`` / ``conv_transpose2d`` / autograd).
"""
import collections

import numpy as np

TIE_MODE = ['all']   # module-level switch for the max-pool gradient routing ('all' | 'first')


def rectify(x):
    return 0.5 * (x + np.abs(x))      # lasagne.nonlinearities.rectify


def identity(x):
    return x


class Param(object):
    def __init__(self, shape, name):
        self.shape = tuple(int(s) for s in shape)
        self.name = name
        self._value = None          # zeros, allocated on first use (shape queries on full-size graphs stay cheap)

    @property
    def value(self):
        if self._value is None:
            self._value = np.zeros(self.shape, dtype=np.float64)
        return self._value

    def set_value(self, v):
        v = np.asarray(v)
        if tuple(v.shape) != self.shape:
            raise ValueError("mismatch: parameter has shape %r but value to set has shape %r" % (self.shape, tuple(v.shape)))
        self._value = v.astype(np.float64)


class Layer(object):
    def __init__(self, incoming, name=None):
        self.input_layer = incoming
        self.input_shape = tuple(incoming.output_shape)
        self.params = collections.OrderedDict()
        self.name = name

    def add_param(self, shape, name):
        p = Param(shape, name)
        self.params[p] = True
        return p

    def get_params(self):
        return list(self.params.keys())

    @property
    def output_shape(self):
        return tuple(self.get_output_shape_for(self.input_shape))

    def get_output_shape_for(self, input_shape):
        return input_shape


class MergeLayer(Layer):
    def __init__(self, incomings, name=None):
        self.input_layers = list(incomings)
        self.input_shapes = [tuple(l.output_shape) if l is not None else None for l in self.input_layers]
        self.params = collections.OrderedDict()
        self.name = name

    @property
    def output_shape(self):
        return tuple(self.get_output_shape_for(self.input_shapes))


class InputLayer(Layer):
    def __init__(self, shape, input_var=None, name=None):
        self.shape = tuple(shape)
        self.input_var = input_var
        self.params = collections.OrderedDict()
        self.name = name

    @property
    def output_shape(self):
        return self.shape


def _pair(v):
    return (int(v), int(v)) if np.isscalar(v) else (int(v[0]), int(v[1]))


class Conv2DLayer(Layer):
    def __init__(self, incoming, num_filters, filter_size, stride=(1, 1), pad=0, untie_biases=False, W=None, b=0,
                 nonlinearity=rectify, flip_filters=True, **kwargs):
        super(Conv2DLayer, self).__init__(incoming, kwargs.get('name'))
        if pad not in (0, (0, 0), 'valid') or untie_biases or not flip_filters:
            raise NotImplementedError("stand-in covers pad='valid', tied biases, flip_filters=True")
        self.num_filters = int(num_filters)
        self.filter_size = _pair(filter_size)
        self.stride = _pair(stride)
        self.nonlinearity = identity if nonlinearity is None else nonlinearity
        self.W = self.add_param((self.num_filters, self.input_shape[1]) + self.filter_size, 'W')
        self.b = None if b is None else self.add_param((self.num_filters,), 'b')

    def get_output_shape_for(self, s):
        fh, fw = self.filter_size
        sy, sx = self.stride
        return (s[0], self.num_filters, (s[2] - fh) // sy + 1, (s[3] - fw) // sx + 1)

    def _taps(self, in_shape):
        fh, fw = self.filter_size
        sy, sx = self.stride
        Ho, Wo = (in_shape[2] - fh) // sy + 1, (in_shape[3] - fw) // sx + 1
        for u in range(fh):
            for v in range(fw):
                r0, c0 = fh - 1 - u, fw - 1 - v
                yield u, v, slice(r0, r0 + sy * (Ho - 1) + 1, sy), slice(c0, c0 + sx * (Wo - 1) + 1, sx)

    def linear(self, x):
        out = np.zeros(self.get_output_shape_for(x.shape), dtype=np.float64)
        W = self.W.value
        for u, v, rows, cols in self._taps(x.shape):
            out += np.einsum('bcyx,oc->boyx', x[:, :, rows, cols], W[:, :, u, v])
        return out

    def forward(self, x):
        y = self.linear(x)
        if self.b is not None:
            y = y + self.b.value.reshape(1, -1, 1, 1)
        return self.nonlinearity(y)

    def vjp(self, x, g):
        if self.nonlinearity is not identity:
            raise NotImplementedError("InverseLayer of a convolution with a nonlinearity")
        gx = np.zeros(x.shape, dtype=np.float64)
        W = self.W.value
        for u, v, rows, cols in self._taps(x.shape):
            gx[:, :, rows, cols] += np.einsum('boyx,oc->bcyx', g, W[:, :, u, v])
        return gx


class BiasLayer(Layer):
    def __init__(self, incoming, b=0, shared_axes='auto', **kwargs):
        super(BiasLayer, self).__init__(incoming, kwargs.get('name'))
        if shared_axes != 'auto':
            raise NotImplementedError
        self.b = self.add_param((self.input_shape[1],), 'b')

    def forward(self, x):
        return x + self.b.value.reshape((1, -1) + (1,) * (x.ndim - 2))


class MaxPool2DLayer(Layer):
    def __init__(self, incoming, pool_size, stride=None, pad=(0, 0), ignore_border=True, **kwargs):
        super(MaxPool2DLayer, self).__init__(incoming, kwargs.get('name'))
        self.pool_size = _pair(pool_size)
        self.stride = self.pool_size if stride is None else _pair(stride)
        if self.stride != self.pool_size or tuple(pad) != (0, 0) or not ignore_border:
            raise NotImplementedError("stand-in covers stride == pool size, no padding, ignore_border=True")

    def get_output_shape_for(self, s):
        return (s[0], s[1], s[2] // self.pool_size[0], s[3] // self.pool_size[1])

    def _windows(self, x):
        ph, pw = self.pool_size
        B, C, H, W = x.shape
        Ho, Wo = H // ph, W // pw
        return x[:, :, :Ho * ph, :Wo * pw].reshape(B, C, Ho, ph, Wo, pw)

    def forward(self, x):
        return self._windows(x).max(axis=(3, 5))

    def vjp(self, x, g):
        ph, pw = self.pool_size
        B, C, H, W = x.shape
        Ho, Wo = H // ph, W // pw
        win = self._windows(x)
        m = win.max(axis=(3, 5), keepdims=True)
        hit = (win == m)
        if TIE_MODE[0] == 'first':
            # first maximum in the window's row-major scan order
            flat = hit.transpose(0, 1, 2, 4, 3, 5).reshape(B, C, Ho, Wo, ph * pw)
            first = np.argmax(flat, axis=-1)
            only = np.zeros_like(flat)
            np.put_along_axis(only, first[..., None], True, axis=-1)
            hit = only.reshape(B, C, Ho, Wo, ph, pw).transpose(0, 1, 2, 4, 3, 5)
        elif TIE_MODE[0] != 'all':
            raise ValueError(TIE_MODE[0])
        gx = np.zeros(x.shape, dtype=np.float64)
        gx[:, :, :Ho * ph, :Wo * pw] = (hit * g[:, :, :, None, :, None]).reshape(B, C, Ho * ph, Wo * pw)
        return gx


class DenseLayer(Layer):
    def __init__(self, incoming, num_units, W=None, b=0, nonlinearity=rectify, **kwargs):
        super(DenseLayer, self).__init__(incoming, kwargs.get('name'))
        self.num_units = int(num_units)
        self.nonlinearity = identity if nonlinearity is None else nonlinearity
        num_inputs = int(np.prod(self.input_shape[1:]))
        self.W = self.add_param((num_inputs, self.num_units), 'W')
        self.b = None if b is None else self.add_param((self.num_units,), 'b')

    def get_output_shape_for(self, s):
        return (s[0], self.num_units)

    def forward(self, x):
        y = x.reshape(x.shape[0], -1) @ self.W.value
        if self.b is not None:
            y = y + self.b.value
        return self.nonlinearity(y)


class ReshapeLayer(Layer):
    def __init__(self, incoming, shape, **kwargs):
        super(ReshapeLayer, self).__init__(incoming, kwargs.get('name'))
        self.shape = tuple(int(s) for s in shape)

    def get_output_shape_for(self, s):
        return self.shape

    def forward(self, x):
        return x.reshape(self.shape)


class InverseLayer(MergeLayer):
    def __init__(self, incoming, layer, **kwargs):
        below = getattr(layer, 'input_layer', None)
        if below is None:
            raise NotImplementedError("InverseLayer of a merge layer")
        super(InverseLayer, self).__init__([incoming, layer, below], kwargs.get('name'))
        self.layer = layer

    def get_output_shape_for(self, shapes):
        return shapes[2]

    def forward(self, inputs):
        incoming, _layer_out, layer_in = inputs
        return self.layer.vjp(layer_in, incoming)


class ConcatLayer(MergeLayer):
    def __init__(self, incomings, axis=1, **kwargs):
        super(ConcatLayer, self).__init__(incomings, kwargs.get('name'))
        self.axis = axis

    def get_output_shape_for(self, shapes):
        out = list(shapes[0])
        out[self.axis] = sum(s[self.axis] for s in shapes)
        return tuple(out)

    def forward(self, inputs):
        return np.concatenate(inputs, axis=self.axis)


class NonlinearityLayer(Layer):
    def __init__(self, incoming, nonlinearity=rectify, **kwargs):
        super(NonlinearityLayer, self).__init__(incoming, kwargs.get('name'))
        self.nonlinearity = identity if nonlinearity is None else nonlinearity

    def forward(self, x):
        return self.nonlinearity(x)


# ------------------------------------------------------------------------------------------------ helpers
def _incomings(layer):
    if hasattr(layer, 'input_layers'):
        return list(layer.input_layers)
    if getattr(layer, 'input_layer', None) is not None:
        return [layer.input_layer]
    return []


def get_all_layers(layer, treat_as_input=None):
    """Topological order of everything ``layer`` depends on: depth-first, incoming layers in the order the layer
    lists them, a layer emitted once all its incomings have been (lasagne.layers.get_all_layers)."""
    roots = list(layer) if isinstance(layer, (list, tuple)) else [layer]
    opened = set(id(l) for l in (treat_as_input or []))
    closed, order = set(), []
    stack = collections.deque(roots)
    while stack:
        top = stack[0]
        if top is None:
            stack.popleft()
        elif id(top) not in opened:
            opened.add(id(top))
            stack.extendleft(reversed(_incomings(top)))
        else:
            stack.popleft()
            if id(top) not in closed:
                closed.add(id(top))
                order.append(top)
    return order


def get_all_params(layer, **tags):
    seen, out = set(), []
    for l in get_all_layers(layer):
        for p in l.get_params():
            if id(p) not in seen:
                seen.add(id(p))
                out.append(p)
    return out


def get_all_param_values(layer, **tags):
    return [p.value for p in get_all_params(layer)]


def set_all_param_values(layer, values, **tags):
    params = get_all_params(layer)
    if len(params) != len(values):
        raise ValueError("mismatch: got %d values to set %d parameters" % (len(values), len(params)))
    for p, v in zip(params, values):
        p.set_value(v)


def get_output(layer_or_layers, inputs=None, **kwargs):
    """Eager evaluation.  ``inputs``: ndarray for the single InputLayer, dict {InputLayer: ndarray}, or None to use
    the ndarray each InputLayer was given as ``input_var``."""
    many = isinstance(layer_or_layers, (list, tuple))
    wanted = list(layer_or_layers) if many else [layer_or_layers]
    values = {}
    for l in get_all_layers(wanted):
        if isinstance(l, InputLayer):
            if isinstance(inputs, dict):
                v = inputs[l]
            elif inputs is not None:
                v = inputs
            else:
                v = l.input_var
            v = np.asarray(v, dtype=np.float64)
            if tuple(v.shape) != tuple(l.shape):
                raise ValueError("input of shape %r for an InputLayer of shape %r" % (tuple(v.shape), tuple(l.shape)))
            values[id(l)] = v
        elif isinstance(l, MergeLayer):
            values[id(l)] = l.forward([values[id(i)] for i in l.input_layers])
        else:
            values[id(l)] = l.forward(values[id(l.input_layer)])
    outs = [values[id(l)] for l in wanted]
    return outs if many else outs[0]


class _Namespace(object):
    pass


layers = _Namespace()
for _n in ('InputLayer', 'Conv2DLayer', 'BiasLayer', 'MaxPool2DLayer', 'DenseLayer', 'ReshapeLayer', 'InverseLayer',
           'ConcatLayer', 'NonlinearityLayer', 'Layer', 'MergeLayer', 'get_all_layers', 'get_all_params',
           'get_all_param_values', 'set_all_param_values', 'get_output'):
    setattr(layers, _n, globals()[_n])
nonlinearities = _Namespace()
nonlinearities.rectify = rectify
nonlinearities.identity = identity
nonlinearities.linear = identity
