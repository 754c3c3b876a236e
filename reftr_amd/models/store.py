"""Flat parameter storage + the nn.Module tree that exposes it under the reference's state_dict names.

All trainable tensors are views of ONE fp32 buffer (`flat_p`), their gradients views of `flat_g`:
  * the optimizer is a single streaming kernel over the buffer (rt_adamw_flat), gradient clipping a single
    reduction (rt_sqnorm), the data-parallel exchange a handful of large all-reduces over `flat_g`;
  * weight-gradient kernels write straight into `flat_g` views: accumulated onto a buffer cleared by one memset, or -- in
    the training loop (`arm_overwrite`) -- the FIRST contribution to a weight matrix overwrites it, so that only the small
    atomically-accumulated tensors (biases, norm parameters, embeddings: the complement of the registered matrices) are
    cleared per step and the matrices' epilogues skip the read of their old value.
Conv weights are stored channels-last ([Cout][kh][kw][Cin], the implicit-GEMM operand order) but exposed
with the reference's logical shape [Cout, Cin, kh, kw] through strides, so state_dict()/load_state_dict()
keep working against reference checkpoints.
"""
import torch
from torch import nn

from . import layout as L


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


def _pad(n, m):
    return (n + m - 1) // m * m


class ParamStore:
    """Owns the flat buffers and the name -> view tables."""

    ALIGN = 4096   # lr-group ranges start on multiples of this (AdamW kernel needs multiples of 4)

    def __init__(self, cfg, device):
        self.cfg = cfg
        self.device = torch.device(device)
        self.table = L.full_table(cfg)
        self.shapes = {n: s for n, s, _ in self.table}
        self.kinds = {n: k for n, _, k in self.table}
        self.physd = L.phys_dims(cfg)      # name -> padded physical shape (RES head convolutions)
        # ---- offsets
        self.offset = {}          # name -> (buffer id, offset)
        self.group_range = {}     # lr group -> (begin, end)
        off = 0
        for grp in (L.GROUP_MAIN, L.GROUP_MASK, L.GROUP_BACKBONE, L.GROUP_BERT):
            begin = off
            for n, s, k in self.table:
                if k == "param" and L.lr_group(n) == grp:
                    ne = _numel(self.physd.get(n, s))
                    assert ne % 4 == 0, (n, s)
                    self.offset[n] = ("p", off)
                    off += ne
            off = _pad(off, self.ALIGN)
            self.group_range[grp] = (begin, off)
        self.n_train = off
        off = 0
        for n, s, k in self.table:
            if k == "frozen":
                self.offset[n] = ("f", off); off += _pad(_numel(s), 4)
        self.n_frozen = off
        off = 0
        for n, s, k in self.table:
            if k == "buffer":
                self.offset[n] = ("b", off); off += _pad(_numel(s), 4)
        self.n_buffer = off
        self.allocate(self.device)

    def allocate(self, device, src=None):
        self.device = torch.device(device)
        from .. import hip as _H
        for tab in (_H._SQACC_MAP, _H._G16_MAP):          # entries of this store's previous buffers (device move): their addresses are void
            for k in [k for k, v in tab.items() if v[0]() is self]:
                del tab[k]
        new = {"p": torch.zeros(self.n_train, dtype=torch.float32, device=device),
               "f": torch.zeros(max(self.n_frozen, 4), dtype=torch.float32, device=device),
               "b": torch.zeros(max(self.n_buffer, 4), dtype=torch.float32, device=device)}
        if src is not None:
            for k in new:
                new[k].copy_(src[k])
        self.flat = new
        self.flat_p = new["p"]
        self.flat_g = torch.zeros(self.n_train, dtype=torch.float32, device=device)
        self.P = {n: self._view(self.flat[b], n, o) for n, (b, o) in self.offset.items()}
        self.G = {n: self._view(self.flat_g, n, o) for n, (b, o) in self.offset.items() if b == "p"}
        # overwrite-mode bookkeeping (see the module docstring): registered matrices {data_ptr: (offset, numel)}
        self._ow, self._ow_table, self._armed, self._written, self._stale_tables = {}, None, False, set(), {}
        self._zero_tab = None
        self.last_stale = ()
        # gradient-norm accumulator (hip._SQACC_MAP, rt_sqnorm_finish): the weight-gradient launches into the registered matrices add
        # |after|^2 - |before|^2 to these slots; `norm_valid` = the slots were cleared together with the gradients (optimizer.zero_grad)
        # and nothing consumed them since.  REFTR_FUSED_NORM=0: the norm is a pass over the buffer again (rt_sqnorm).
        import os
        from .. import hip as H
        self.fused_norm = os.environ.get("REFTR_FUSED_NORM", "1") != "0" and self.device.type == "cuda"
        self.sq_slots = torch.zeros(H.SQ_SLOTS * H.SQ_STRIDE, dtype=torch.float32, device=device) if self.fused_norm else None
        self.norm_valid = False

    # ------------------------------------------------------------------ overwrite-mode gradient production
    def register_overwritable(self, gw):
        """`gw`: a contiguous piece of flat_g (a weight matrix's gradient) whose every producer asks `claim` first."""
        assert gw.is_contiguous() and gw.dtype == torch.float32
        off = (gw.data_ptr() - self.flat_g.data_ptr()) // 4
        assert 0 <= off and off + gw.numel() <= self.flat_g.numel()
        self._ow[gw.data_ptr()] = (off, gw.numel())
        self._ow_table, self._stale_tables, self._zero_tab = None, {}, None
        if self.fused_norm:
            from .. import hip as H
            import weakref
            H._SQACC_MAP[gw.data_ptr()] = (weakref.ref(self), self.sq_slots)

    def _complement_table(self):
        """Static device table of <= 16384-element chunks covering everything that is NOT a registered matrix."""
        if self._ow_table is None:
            chunks, pos = [], 0
            for off, n in sorted(self._ow.values()) + [(self.flat_g.numel(), 0)]:
                a = pos
                while a < off:
                    c = min(16384, off - a)
                    chunks += [a, c]; a += c
                pos = max(pos, off + n)
            self._ow_table = (torch.tensor(chunks, dtype=torch.int64, device=self.device), len(chunks) // 2)
        return self._ow_table

    def _zero_table(self):
        """The complement table plus the clip-norm accumulator slots (addressed relative to the gradient buffer: rt_zero_chunks takes
        64-bit element offsets, both allocations are 512-byte aligned): ONE launch clears everything a backward accumulates into."""
        # the table bakes sq_slots' address RELATIVE to flat_g in: it is only valid for this pair of allocations
        key = (self.flat_g.data_ptr(), self.sq_slots.data_ptr() if self.fused_norm else 0)
        if self._zero_tab is not None and getattr(self, "_zero_tab_key", None) != key:
            self._zero_tab = None
        if self._zero_tab is None:
            self._zero_tab_key = key
            table, n = self._complement_table()
            extra = []
            if self.fused_norm:
                d = self.sq_slots.data_ptr() - self.flat_g.data_ptr()
                assert d % 16 == 0
                a, tot = 0, self.sq_slots.numel()
                while a < tot:
                    c = min(16384, tot - a)
                    extra += [d // 4 + a, c]; a += c
            if extra:
                table = torch.cat([table, torch.tensor(extra, dtype=torch.int64, device=self.device)])
            self._zero_tab = (table, n + len(extra) // 2)
        return self._zero_tab

    def zero_for_backward(self, fast=True):
        """What optimizer.zero_grad does to the gradient buffer in front of a backward, plus the clip-norm accumulator.  fast: only
        the atomically-accumulated tensors are cleared (the weight matrices are overwritten by their first producer) and the
        accumulator slots go in the same launch; REFTR_OVERWRITE=0 or a CPU store: the full clear.  A method of the store so that
        the model can issue it on the language stream at the forward join, off the loss -> backward chain.  Both branches end with
        the norm accumulator cleared and marked valid."""
        import os
        if fast and os.environ.get("REFTR_OVERWRITE", "1") != "0" and self.flat_g.is_cuda:
            self.arm_overwrite()              # one launch: the accumulated tensors AND the norm accumulator slots
            if self.fused_norm:
                self.norm_valid = True
        else:
            self.disarm()                     # a backward that was abandoned half-way must not leave overwrite mode armed
            self.flat_g.zero_()
            g16 = getattr(self, "flat_g16", None)
            if g16 is not None:
                # full-clear mode under the bf16 exchange: a matrix that no producer writes in this step has no producer for its
                # twin either (the rounding pass of a slice skips registered matrices), and finish_overwrite(_range) only repairs
                # twins while armed -- so the twins are cleared with the masters (the previous step's all-reduced values would
                # otherwise be exchanged and applied again)
                g16.zero_()
            self.begin_norm()                 # the producers' epilogues collect the clip norm from here on (rt_sqnorm_finish)

    def begin_norm(self):
        """The gradients are (about to be) cleared / re-armed: clear the norm accumulator with them."""
        if self.fused_norm:
            self.sq_slots.zero_()
            self.norm_valid = True

    def arm_overwrite(self):
        """Start of a training backward: clear the atomically-accumulated tensors only; until `finish_overwrite` the first
        `claim` of every registered matrix answers True (its producer overwrites)."""
        from .. import hip as H
        table, n = self._zero_table()
        if n:
            H.zero_chunks(self.flat_g, table, n)
        self._armed, self._written = True, set()
        self._range_stale = ()

    def claim(self, gw):
        if not self._armed:
            return False
        key = gw.data_ptr()
        if key not in self._ow or key in self._written:
            return False
        self._written.add(key)
        return True

    def disarm(self):
        """Forget a backward that never reached `finish_overwrite` (an exception, an abandoned data-parallel phase generator):
        the caller clears the whole buffer, nothing may stay in overwrite mode."""
        self._armed, self._written = False, set()

    def finish_overwrite_range(self, bounds):
        """Data parallel: the slices `bounds` = [(a, b), ...] of the gradient buffer are final NOW -- their exchange is the next thing
        that happens -- so the registered matrices inside them that no producer has written in this step are cleared here, in front
        of the exchange, instead of at the end of backward (where the clear would come after the slice was copied / all-reduced:
        the previous step's gradient of such a matrix would be exchanged and applied, and with the fp32 exchange the late clear
        would race the in-place all-reduce).  They count as written from here on: finish_overwrite leaves them alone."""
        if not self._armed:
            return
        merged = []
        for a, b in sorted((a, b) for a, b in bounds if b > a):
            if merged and a <= merged[-1][1]:
                merged[-1] = (merged[-1][0], max(merged[-1][1], b))
            else:
                merged.append((a, b))
        missing = frozenset(k for k, (off, n) in self._ow.items()
                            if k not in self._written and any(a <= off < b for a, b in merged))
        if not missing:
            return
        self._zero_matrices(missing)
        self._written |= missing
        self._range_stale = getattr(self, "_range_stale", ()) + tuple(self._ow[k] for k in missing)

    def _zero_matrices(self, keys):
        ent = self._stale_tables.get(keys)
        if ent is None:
            chunks = []
            for off, n in sorted(self._ow[k] for k in keys):
                a = off
                while a < off + n:
                    c = min(16384, off + n - a)
                    chunks += [a, c]; a += c
            ent = self._stale_tables[keys] = (torch.tensor(chunks, dtype=torch.int64, device=self.device), len(chunks) // 2)
        from .. import hip as H
        H.zero_chunks(self.flat_g, ent[0], ent[1])
        g16 = getattr(self, "flat_g16", None)
        if g16 is not None and g16.is_cuda:          # the bf16 exchange twins of the cleared matrices (their producers never ran)
            H.round_chunks(self.flat_g, g16, ent[0], ent[1])

    def finish_overwrite(self):
        """End of backward: EVERY registered matrix that no producer wrote in this step is cleared -- unconditionally, by one
        launch that depends on this step's written set only, never on what earlier steps did.  (A matrix that an input kind
        never produces -- the decoder's self-attention q/k projections with one query per image, map_phrase's second pass --
        would otherwise keep the gradient of whichever step last wrote it; a captured graph bakes its host-side decisions in,
        so the clear must be part of every step's own launches: a T == 1 graph replayed after a T > 1 graph clears what the
        T > 1 step left behind.)"""
        self.last_stale = ()
        if not self._armed:
            return
        missing = frozenset(self._ow) - self._written
        # (offset, numel) of what this step left unwritten (including what finish_overwrite_range already cleared at a boundary)
        self.last_stale = tuple(self._ow[k] for k in missing) + getattr(self, "_range_stale", ())
        self._range_stale = ()
        if missing:
            self._zero_matrices(missing)
        self._armed, self._written = False, set()

    def _view(self, buf, name, off):
        shape = self.shapes[name]
        if name in self.physd:             # padded storage: logical tensor = unpadded corner of the physical one
            ps = self.physd[name]
            flat = buf[off:off + _numel(ps)]
            if len(ps) == 4:
                co, ci, kh, kw = shape
                return flat.view(ps)[:co, :, :, :ci].permute(0, 3, 1, 2)
            return flat.view(ps)[:shape[0]]
        flat = buf[off:off + _numel(shape)]
        if len(shape) == 4:     # conv weight: physical [Cout][kh][kw][Cin], logical [Cout, Cin, kh, kw]
            co, ci, kh, kw = shape
            return flat.view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return flat.view(shape)

    def view_of(self, buf, name):
        """The logical view of tensor `name` inside another flat buffer laid out like flat_p (optimizer moments)."""
        b, off = self.offset[name]
        assert b == "p"
        return self._view(buf, name, off)

    def phys(self, name, grad=False):
        """Physical (contiguous) tensor of a conv weight / its gradient: [Cout, kh*kw, Cin]."""
        if name in self.physd:             # padded: the whole physical block [Cout_pad, kh*kw, Cin_pad] / [Cout_pad]
            ps = self.physd[name]
            b, off = self.offset[name]
            buf = self.flat_g if grad else self.flat[b]
            flat = buf[off:off + _numel(ps)]
            return flat.view(ps[0], ps[1] * ps[2], ps[3]) if len(ps) == 4 else flat
        t = (self.G if grad else self.P)[name]
        co, ci, kh, kw = self.shapes[name]
        return t.permute(0, 2, 3, 1).reshape(co, kh * kw, ci)    # a view: permute back to storage order

    def packed(self, first, n_parts, grad=False):
        """View of `n_parts` adjacent, equally-shaped tensors starting at `first` as one stacked tensor."""
        b, off = self.offset[first]
        shape = self.shapes[first]
        buf = self.flat_g if grad else self.flat[b]
        n = _numel(shape)
        return buf[off:off + n_parts * n].view((n_parts * shape[0],) + tuple(shape[1:]))


def build_module_tree(root: nn.Module, store: ParamStore):
    """Registers every tensor of the store on `root` under its dotted reference name (containers are plain
    nn.Module objects), so named_parameters()/state_dict() produce the reference's keys."""
    for name, shape, kind in store.table:
        parts = name.split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        leaf = parts[-1]
        if kind == "buffer":
            mod.register_buffer(leaf, store.P[name])
        else:
            par = nn.Parameter(store.P[name], requires_grad=(kind == "param"))
            if kind == "param":
                par.grad = store.G[name]
            mod.register_parameter(leaf, par)


def rebind(root: nn.Module, store: ParamStore):
    """After the flat buffers moved (device change), point every Parameter / buffer at the new views."""
    for name, shape, kind in store.table:
        parts = name.split(".")
        mod = root
        for p in parts[:-1]:
            mod = mod._modules[p]
        leaf = parts[-1]
        if kind == "buffer":
            mod._buffers[leaf] = store.P[name]
        else:
            par = mod._parameters[leaf]
            par.data = store.P[name]
            if kind == "param":
                par.grad = store.G[name]
