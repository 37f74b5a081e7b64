"""FusedSGD -- torch.optim.SGD semantics as the reference configures it (train.py:80:
lr 1e-3, momentum 0.9, weight_decay 1e-4, dampening 0, no nesterov), stepped on ONE flat fp32
buffer by one libsfvos kernel (sfvos_sgd_step).

The parameters stay ordinary nn.Parameters (the reference's `SGD(model.parameters())`
keeps working on them); FusedSGD re-homes their storage into a flat buffer so that
  * the optimiser step is a single launch, and
  * the data-parallel gradient all-reduce is a single RCCL collective on the flat grad."""
import ctypes

import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.9, weight_decay=1e-4):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError('FusedSGD got no trainable parameters')
        super(FusedSGD, self).__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._params = params
        dev = params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in params):
            raise ValueError('FusedSGD needs all parameters fp32 on one device')
        n = sum(p.numel() for p in params)
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_buf = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        self._span = {}   # id(param) -> [lo, hi) of the flat buffers
        self.bucket = None
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self._span[id(p)] = (off, off + k)
                self.flat_param[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_param[off:off + k].view_as(p)
                p.grad = self.flat_grad[off:off + k].view_as(p)
                off += k
        self._steps = 0
        self._views = {id(p): p.grad for p in params}
        self._clean = False  # True between zero_grad() and the first gradient written after it

    def attach(self, module, bucket=None):
        """Let `module` (a SlowFastLayers) write its parameter gradients straight into the flat gradient buffer
        (overwrite on the first backward after zero_grad(), accumulate in place on later ones): no per-parameter
        temporaries and no `grad += tmp` launches.  dgamma/dbeta of a BatchNorm share one kernel, so both or neither
        of them must be FusedSGD parameters (always true for SlowFastLayers.parameters()).
        bucket: a GradBucket over self.flat_grad; when it is armed, every layer's slice of the flat gradient is
        all-reduced as soon as backward has produced it (overlap of the exchange with the rest of backward)."""
        module._grad_sink = self
        self.bucket = bucket
        if bucket is not None and hasattr(bucket, 'set_buckets'):
            # coalesced exchange: (layer 3), (layer 2), (both laterals), (layer 1) -- each group is one contiguous
            # range of the flat buffer in the reference's registration order (model.py:47-67: f1 s1 f2 s2 f3 s3 l1 l2)
            # and completes in that order during backward (f3 s3 l2 f2 s2 l1 f1 s1): layer 3 goes out first and only
            # layer 1's bucket has no backward left to overlap with (with the laterals in layer 3's bucket
            # nothing could go out before conv_f2s1's backward)
            ranges = []
            for group in (('f3', 's3'), ('f2', 's2'), ('l1', 'l2'), ('f1', 's1')):
                spans = []
                for name in group:
                    l = module.plan.layer(name)
                    for mod in (getattr(module, l.conv), getattr(module, l.bn)):
                        spans += [self._span[id(p)] for p in mod.parameters() if id(p) in self._span]
                if not spans:
                    continue
                lo, hi = min(a for a, _ in spans), max(b for _, b in spans)
                if hi - lo == sum(b - a for a, b in spans):
                    ranges.append((lo, hi))
            bucket.set_buckets(ranges)
        return self

    def layer_done(self, params, stream=None):
        """Called by the module's backward when every gradient of `params` (one layer: conv + its BatchNorm) has
        been enqueued on `stream`: hands the smallest covering range of the flat gradient to the armed bucket."""
        b = getattr(self, 'bucket', None)
        if b is None or not b.armed:
            return
        spans = [self._span[id(p)] for p in params if id(p) in self._span]
        if not spans:
            return
        lo, hi = min(s[0] for s in spans), max(s[1] for s in spans)
        if hi - lo != sum(s[1] - s[0] for s in spans):
            return   # not contiguous in the flat buffer: left for GradBucket.finish()
        b.segment_ready(lo, hi, stream)

    # -- gradient-sink protocol used by SlowFastLayers._engine_backward
    def begin_direct(self, device):
        """0: not available; 1: overwrite (first backward after zero_grad()); 2: accumulate."""
        if device != self.flat_grad.device:
            return 0
        for p in self._params:  # someone re-pointed a .grad: fall back to autograd accumulation
            if p.grad is not self._views[id(p)]:
                return 0
        return 1 if self._clean else 2

    def view_of(self, p):
        return self._views.get(id(p))

    def end_direct(self):
        self._clean = False

    # -- checkpoint interchange with torch.optim.SGD (reference train.py:117-121 saves opt.state_dict(),
    # train.py:57-60 / train_osvos.py restore it): same dictionary layout in both directions
    def state_dict(self):
        """torch.optim.SGD's state-dict layout: one param group with torch's own hyper-parameter keys and, once a
        step has been taken, a `momentum_buffer` per parameter (copies of the slices of the flat buffer)."""
        template = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-3).state_dict()['param_groups'][0]
        g = self.param_groups[0]
        group = dict(template)
        group.update(lr=g['lr'], momentum=g['momentum'], weight_decay=g['weight_decay'], dampening=0, nesterov=False,
                     params=list(range(len(self._params))))
        state = {}
        if self._steps > 0:
            off = 0
            for i, p in enumerate(self._params):
                k = p.numel()
                state[i] = {'momentum_buffer': self.flat_buf[off:off + k].view_as(p).clone()}
                off += k
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, state_dict):
        """Accepts what FusedSGD.state_dict() or torch.optim.SGD(...).state_dict() produced for the same
        parameter list (dampening 0, no nesterov -- the reference's configuration)."""
        groups = state_dict['param_groups']
        if len(groups) != 1 or len(groups[0]['params']) != len(self._params):
            raise ValueError('FusedSGD.load_state_dict: expected one param group with %d parameters' % len(self._params))
        g = groups[0]
        if g.get('dampening', 0) != 0 or g.get('nesterov', False) or g.get('maximize', False):
            raise ValueError('FusedSGD implements dampening 0, no nesterov, no maximize')
        self.param_groups[0].update(lr=g['lr'], momentum=g['momentum'], weight_decay=g['weight_decay'])
        state = state_dict.get('state', {})
        have = [i for i in range(len(self._params)) if i in state and state[i].get('momentum_buffer') is not None]
        if have and len(have) != len(self._params):
            raise ValueError('FusedSGD.load_state_dict: momentum buffers for only some of the parameters')
        with torch.no_grad():
            if have:
                off = 0
                for i, p in enumerate(self._params):
                    k = p.numel()
                    buf = state[i]['momentum_buffer']
                    if buf.numel() != k:
                        raise ValueError('FusedSGD.load_state_dict: momentum buffer %d has the wrong size' % i)
                    self.flat_buf[off:off + k].copy_(buf.reshape(-1).to(self.flat_buf.device, torch.float32))
                    off += k
                self._steps = max(self._steps, 1)   # next step uses momentum * buf + g
            else:
                self.flat_buf.zero_()
                self._steps = 0                     # next step initialises the buffers with g, as torch does

    def zero_grad(self, set_to_none=False):
        # grads are views of flat_grad: zero in place, never detach them
        self.flat_grad.zero_()
        for p in self._params:
            if p.grad is not self._views[id(p)]:
                p.grad = self._views[id(p)]
        self._clean = True

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise RuntimeError('FusedSGD does not take a closure')
        g = self.param_groups[0]
        if not self.flat_param.is_cuda:
            raise RuntimeError('FusedSGD runs only on the GPU through libsfvos.so (no CPU fallback)')
        for p in self._params:  # a caller may have replaced .grad (e.g. set_to_none); fold it back
            if p.grad is None:
                continue
            if p.grad.data_ptr() < self.flat_grad.data_ptr() or \
                    p.grad.data_ptr() >= self.flat_grad.data_ptr() + self.flat_grad.numel() * 4:
                raise RuntimeError('FusedSGD: a parameter .grad was re-allocated outside the flat gradient buffer; '
                                   'use optimizer.zero_grad() (set_to_none=False)')
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.call('sfvos_sgd_step', _ptr(self.flat_param), _ptr(self.flat_grad), _ptr(self.flat_buf),
                  self.flat_param.numel(), float(g['lr']), float(g['momentum']), float(g['weight_decay']),
                  1 if self._steps == 0 else 0, st)
        self._steps += 1
        # the kernel wrote the parameters through raw pointers: autograd's version counters did not
        # move, so invalidate the packed-weight caches explicitly
        _lib.bump_weight_epoch()
