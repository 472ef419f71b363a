"""Tensor-level wrappers over the C ABI: check tensors, pull data_ptr()/current stream, call libreco_hip.

Everything here runs on the GPU through libreco_hip.so; there is no eager-PyTorch arithmetic on the
product path.  torch is used for allocation (torch.empty), streams and device guards only.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ACT_CLAMP, ACT_NONE, ACT_RATING, ACT_RELU, ACT_SIGMOID, TowerDesc, check  # noqa: F401

ACTS = {None: ACT_NONE, "none": ACT_NONE, "relu": ACT_RELU, "sigmoid": ACT_SIGMOID, "clamp": ACT_CLAMP, "rating": ACT_RATING}


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("recommender_amd needs an AMD GPU (gfx950 / MI355X): there is no CPU fallback")


def _f32(t, name, dim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    if dim is not None and t.dim() != dim:
        raise ValueError("%s must be %d-D, got shape %s" % (name, dim, tuple(t.shape)))
    return t if t.is_contiguous() else t.contiguous()


def _i64(t, name, device):
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t, dtype=torch.int64)
    if t.dtype != torch.int64:
        if t.dtype in (torch.int32, torch.int16, torch.int8, torch.uint8):
            t = t.to(torch.int64)
        else:
            raise TypeError("%s must be an integer tensor, got %s" % (name, t.dtype))
    if t.dim() != 1:
        raise ValueError("%s must be 1-D, got shape %s" % (name, tuple(t.shape)))
    t = t.to(device, non_blocking=True)
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


# ----------------------------------------------------------------------------------------------
# K-a gathers
# ----------------------------------------------------------------------------------------------
def gather_rows(table, idx, check_ids=True):
    """out[b] = table[idx[b]]  (nn.Embedding forward, ncf_model.py:207-208)."""
    table = _f32(table, "table", 2)
    idx = _i64(idx, "idx", table.device)
    out = torch.empty((idx.numel(), table.shape[1]), dtype=torch.float32, device=table.device)
    oob = torch.zeros(1, dtype=torch.int32, device=table.device) if check_ids else None
    with torch.cuda.device(table.device):
        check(_lib.lib().reco_gather_rows(_p(table), table.shape[0], table.shape[1], _p(idx), idx.numel(), _p(out),
                                          _p(oob), _stream(table.device)))
    if check_ids and int(oob.item()):
        raise IndexError("index out of range in self")
    return out


def neumf_gather(eu_g, ei_g, eu_m, ei_m, user, item, check_ids=True, gmf_out=None):
    """(gmf_prod [B,dg], mlp_in [B,2*dm]) as the reference materialises them (ncf_model.py:207-224).

    gmf_out: optional [B, >=dg] float32 view with unit inner stride (a column block of a wider buffer) to receive gmf_prod.
    """
    dev = (eu_g if eu_g is not None else eu_m).device
    user = _i64(user, "user", dev); item = _i64(item, "item", dev)
    if user.numel() != item.numel():
        raise ValueError("user and item must have the same length")
    B = user.numel()
    gp = mi = None
    n_users = n_items = dg = dm = 0
    if eu_g is not None:
        eu_g = _f32(eu_g, "embed_user_GMF", 2); ei_g = _f32(ei_g, "embed_item_GMF", 2)
        n_users, dg = eu_g.shape; n_items = ei_g.shape[0]
        gp = gmf_out if gmf_out is not None else torch.empty((B, dg), dtype=torch.float32, device=dev)
    if eu_m is not None:
        eu_m = _f32(eu_m, "embed_user_MLP", 2); ei_m = _f32(ei_m, "embed_item_MLP", 2)
        n_users, dm = eu_m.shape; n_items = ei_m.shape[0]
        mi = torch.empty((B, 2 * dm), dtype=torch.float32, device=dev)
    oob = torch.zeros(1, dtype=torch.int32, device=dev) if check_ids else None
    with torch.cuda.device(dev):
        check(_lib.lib().reco_neumf_gather(_p(eu_g), _p(ei_g), _p(eu_m), _p(ei_m), n_users, n_items, dg, dm,
                                           _p(user), _p(item), B, _p(gp), gp.stride(0) if gp is not None and B > 0 else 0,
                                           _p(mi), 0, _p(oob), _stream(dev)))
    if check_ids and int(oob.item()):
        raise IndexError("index out of range in self")
    return gp, mi


# ----------------------------------------------------------------------------------------------
# K-c Linear / AutoRec
# ----------------------------------------------------------------------------------------------
def linear(x, weight, bias=None, act=None, lo=0.0, hi=0.0, out=None, x_transposed=False, bias_per_row=False, precision=None):
    """act(x @ weight.T + bias) on the fp32 MFMA GEMM (nn.Linear + fused activation).

    x_transposed: x is given as [K, M] (the kernel reads it transposed: no .t().contiguous() copy);
    bias_per_row: bias has M entries and is added per output row;
    precision: "f32" / "bf16x3" (None: the process-wide setting, see set_tower_precision).
    """
    if isinstance(x, torch.Tensor) and x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and x.stride(1) == 1 \
            and x.stride(0) >= x.shape[1]:
        pass                                        # a column block of a wider row-major buffer is fine: lda = stride(0)
    else:
        x = _f32(x, "input", 2)
    weight = _f32(weight, "weight", 2)
    if bias is not None:
        bias = _f32(bias, "bias", 1)
    K, M = x.shape if x_transposed else (x.shape[1], x.shape[0])
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError("mat1 and mat2 shapes cannot be multiplied (%dx%d and %dx%d)" % (M, K, weight.shape[1], N))
    if bias is not None and bias.numel() != (M if bias_per_row else N):
        raise ValueError("bias has %d entries, expected %d" % (bias.numel(), M if bias_per_row else N))
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    elif out.shape != (M, N) or out.dtype != torch.float32 or (N > 1 and out.stride(1) != 1):
        raise ValueError("out must be a float32 [%d, %d] tensor with unit inner stride" % (M, N))
    L = _lib.lib()
    nbytes = L.reco_linear_workspace_bytes(M, N, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    lda = x.stride(0) if x.shape[0] > 1 else x.shape[1]
    if precision is not None and precision not in PRECISIONS:
        raise ValueError("precision must be one of %s, got %r" % (sorted(PRECISIONS), precision))
    flags = (1 if x_transposed else 0) | (2 if bias_per_row else 0) | (4 if PRECISIONS[precision or tower_precision()] else 0)
    with torch.cuda.device(x.device):
        check(L.reco_linear_f32(_p(x), lda, None, 0, M, _p(weight), K, _p(bias), N, K, ACTS[act], lo, hi,
                                _p(out), out.stride(0) if M > 1 else N, flags, _p(ws), nbytes, _stream(x.device)))
    return out


def mlp2_supported(k1, n2, n3):
    return bool(_lib.lib().reco_mlp2_supported(int(k1), int(n2), int(n3)))


def mlp2(x, w2, b2, w3, b3, act2="relu", act3=None, lo=0.0, hi=0.0):
    """act3(act2(x @ w2.T + b2) @ w3.T + b3) in one launch (reco_mlp2_f32): two narrow Linear layers whose intermediate stays in
    LDS.  Same bits as two linear() calls.  Check mlp2_supported(K1, N2, N3) first."""
    if not (isinstance(x, torch.Tensor) and x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and x.stride(1) == 1
            and x.stride(0) >= x.shape[1] and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0):
        x = _f32(x, "input", 2)
    w2 = _f32(w2, "w2", 2); w3 = _f32(w3, "w3", 2)
    b2 = None if b2 is None else _f32(b2, "b2", 1)
    b3 = None if b3 is None else _f32(b3, "b3", 1)
    M, K1 = x.shape
    N2, N3 = w2.shape[0], w3.shape[0]
    if w2.shape[1] != K1 or w3.shape[1] != N2 or (b2 is not None and b2.numel() != N2) or (b3 is not None and b3.numel() != N3):
        raise ValueError("mlp2: shapes do not chain: x %s w2 %s w3 %s" % (tuple(x.shape), tuple(w2.shape), tuple(w3.shape)))
    out = torch.empty((M, N3), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().reco_mlp2_f32(_p(x), x.stride(0) if M > 1 else K1, M, K1, _p(w2), _p(b2), N2, ACTS[act2], _p(w3), _p(b3), N3, ACTS[act3],
                                       lo, hi, _p(out), N3, _stream(x.device)))
    return out


def transpose(x):
    """x.T as a new contiguous tensor (HIP kernel; used for the [K, N] weight layout of csr_linear)."""
    x = _f32(x, "input", 2)
    out = torch.empty((x.shape[1], x.shape[0]), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().reco_transpose_f32(_p(x), x.shape[0], x.shape[1], x.shape[1], _p(out), x.shape[0], _stream(x.device)))
    return out


def csr_k_ranges(K, N):
    """How many k ranges (XCD placement, CsrMatrix) pay for a [K, N] fp32 weight matrix: none while it is small enough to sit in
    every XCD's 4-MB L2 next to the Infinity Cache (the Hybrid encoders' 3.8 / 6.2 MB: 1 range 0.080 / 0.114 ms against 0.106 /
    0.153 ms with 8), four for the 12-MB AutoRec encoder (0.155 ms against 0.231 with one and 0.173 with eight), eight beyond.
    Measured on an ML-1M-like matrix with Zipf item popularity (tools/csr_exp.py); more ranges also mean more, shorter segments."""
    nbytes = int(K) * int(N) * 4
    return 1 if nbytes <= (8 << 20) else (4 if nbytes <= (16 << 20) else 8)


class CsrMatrix:
    """A float32 matrix kept SPARSE on the device: CSR arrays plus the segment table of the sparse Linear kernel.

    Built once per rating matrix (start-up work, like packing weights), from a dense tensor or from (row, col, value)
    triplets; the scoring path then never touches the zeros.  `shape` = (rows, cols).

    Segment table (include/reco_hip.h, reco_csr_linear_f32): the columns are cut into `k_ranges` ranges that hold equal shares
    of the nonzeros; a segment is at most SEG_LEN (skewed matrices: SEG_LEN_SKEWED) consecutive nonzeros of one row inside one range; `order` places segment
    lists so that workgroup b works in range b % k_ranges (one eighth of the weight matrix per XCD on MI355X).
    """
    SEG_LEN = 256            # segment cap while every (row, range) group fits in one segment
    SEG_LEN_SKEWED = 96      # ... and when some do not (popular items): see _build_segments
    K_RANGES = 8
    SEGS_PER_WG = 4          # RECO_CSR_SEGS_PER_WG

    def __init__(self, row_ptr, col, val, shape, k_ranges=None):
        dev = val.device
        if not val.is_cuda:
            raise ValueError("CsrMatrix lives on the GPU")
        self.shape = (int(shape[0]), int(shape[1]))
        self.row_ptr = row_ptr.to(device=dev, dtype=torch.int64).contiguous()
        self.col = col.to(device=dev, dtype=torch.int32).contiguous()
        self.val = val.to(dtype=torch.float32).contiguous()
        if self.row_ptr.numel() != self.shape[0] + 1 or self.col.numel() != self.val.numel():
            raise ValueError("CsrMatrix: row_ptr has %d entries for %d rows; col / val have %d / %d entries"
                             % (self.row_ptr.numel(), self.shape[0], self.col.numel(), self.val.numel()))
        self.nnz = int(self.val.numel())
        self.device = dev
        self.k_ranges = int(self.K_RANGES if k_ranges is None else k_ranges)
        self._build_segments()

    def _build_segments(self):
        dev, (M, K), R, SEG = self.device, self.shape, max(1, self.k_ranges), self.SEG_LEN
        i64 = dict(dtype=torch.int64, device=dev)
        cnt = self.row_ptr[1:] - self.row_ptr[:-1]
        col = self.col.long()
        if R > 1 and self.nnz > 0:                      # k ranges with equal shares of the nonzeros
            cum = torch.cumsum(torch.bincount(col.clamp(0, K - 1), minlength=K), 0)
            targets = (torch.arange(1, R, **i64) * self.nnz) // R
            bounds = torch.searchsorted(cum, targets) + 1                  # range r = [bounds[r-1], bounds[r])
            rng = torch.bucketize(col, bounds, right=True)
        else:
            R = 1
            rng = torch.zeros(self.nnz, **i64)
        nz_row = torch.repeat_interleave(torch.arange(M, **i64), cnt)
        gcnt = torch.bincount(nz_row * R + rng, minlength=M * R)           # nonzeros per (row, range); contiguous in CSR order
        gstart = torch.cumsum(gcnt, 0) - gcnt
        # A wave walks its segment one nonzero at a time (a weight row per step), so the LONGEST segment bounds the launch and the
        # segments of a popular row run side by side on different waves.  While every group fits in SEG_LEN nothing is cut (with
        # one range every row is one segment: no partial sums, no reduce launch, the dense layer's own fmaf chain).  Once some
        # groups are longer -- item-major rows of a Zipf-popular catalogue: 5,700 raters against 270 on average -- shorter
        # segments pack better than fewer partial sums save (tools/csr_seg_exp.py, csr + reduce ms at caps 256 / 128 / 96 / 64:
        # AutoRec encoder 0.150 / 0.127 / 0.125 / 0.125, Hybrid item side 0.115 / 0.098 / 0.096 / 0.099; longer caps lose badly:
        # 512 0.179 / 0.199, one segment per row 1.05 ms).
        if self.nnz > 0 and int(gcnt.max()) > SEG:
            SEG = self.SEG_LEN_SKEWED
        self.seg_len = SEG
        gsegs = (gcnt + SEG - 1) // SEG                                    # no segment for an empty (row, range)
        gfirst = torch.cumsum(gsegs, 0) - gsegs
        self.n_seg = int(gsegs.sum())
        group = torch.repeat_interleave(torch.arange(M * R, **i64), gsegs)
        within = torch.arange(self.n_seg, **i64) - gfirst[group]
        self.seg_begin = (gstart[group] + within * SEG).contiguous()
        self.seg_end = torch.minimum(self.seg_begin + SEG, gstart[group] + gcnt[group]).contiguous()
        self.seg_row = (group // R).to(torch.int32).contiguous()
        seg_rng = group % R
        row_segs = gsegs.view(M, R).sum(1)
        self.row_seg_ptr = torch.zeros(M + 1, **i64)
        self.row_seg_ptr[1:] = torch.cumsum(row_segs, 0)
        self.single_segment_rows = bool(M > 0 and int((row_segs == 1).all()))
        self.order = None
        if R > 1 and self.n_seg > 0:
            # placement table: SPW slots per workgroup; the i-th segment of range r goes to workgroup (i // SPW) * R + r
            SPW = self.SEGS_PER_WG
            per = torch.bincount(seg_rng, minlength=R)
            by_range = torch.argsort(seg_rng, stable=True)
            r = seg_rng[by_range]
            pos = torch.arange(self.n_seg, **i64) - (torch.cumsum(per, 0) - per)[r]
            groups = (int(per.max()) + SPW - 1) // SPW
            order = torch.full((groups * R * SPW,), -1, dtype=torch.int32, device=dev)
            order[((pos // SPW) * R + r) * SPW + pos % SPW] = by_range.to(torch.int32)
            self.order = order.contiguous()

    @classmethod
    def from_dense(cls, m, transpose=False, k_ranges=None):
        """CSR of m (or of m.T) for a dense [rows, cols] device tensor; entries != 0 are kept."""
        m = _f32(m, "matrix", 2)
        v = m.t() if transpose else m
        idx = v.nonzero()                                   # sorted by (row, col)
        counts = torch.bincount(idx[:, 0], minlength=v.shape[0])
        row_ptr = torch.zeros(v.shape[0] + 1, dtype=torch.int64, device=m.device)
        row_ptr[1:] = torch.cumsum(counts, 0)
        return cls(row_ptr, idx[:, 1], v[idx[:, 0], idx[:, 1]], v.shape, k_ranges)

    @classmethod
    def from_triplets(cls, rows, cols, vals, shape, device):
        """CSR from host (row, col, value) triplets; for a repeated (row, col) the LAST occurrence wins
        (data_loader.py:96-103 assigns in file order), zeros are dropped."""
        from .data_io import csr_arrays
        row_ptr, c, v = csr_arrays(rows, cols, vals, shape[0], shape[1])
        dev = torch.device(device)
        return cls(torch.from_numpy(row_ptr).to(dev), torch.from_numpy(c).to(dev), torch.from_numpy(v).to(dev), shape)

    def to_dense(self):
        out = torch.zeros(self.shape, dtype=torch.float32, device=self.device)
        rows = torch.repeat_interleave(torch.arange(self.shape[0], device=self.device), self.row_ptr[1:] - self.row_ptr[:-1])
        out[rows, self.col.long()] = self.val
        return out


def csr_linear(x, weight_t, bias=None, act=None, lo=0.0, hi=0.0, out=None):
    """act(x @ weight_t + bias) for a CsrMatrix x [M, K] and weight_t = W.T [K, N] (see `transpose`): only the nonzeros of x
    are visited (reco_csr_linear_f32)."""
    if not isinstance(x, CsrMatrix):
        raise TypeError("x must be a CsrMatrix")
    weight_t = _f32(weight_t, "weight_t", 2)
    M, K = x.shape
    N = weight_t.shape[1]
    if weight_t.shape[0] != K or weight_t.device != x.device:
        raise ValueError("weight_t must be a [%d, N] float32 tensor on %s, got %s on %s" % (K, x.device, tuple(weight_t.shape), weight_t.device))
    if bias is not None:
        bias = _f32(bias, "bias", 1)
        if bias.numel() != N or bias.device != x.device:
            raise ValueError("bias has %d entries, expected %d" % (bias.numel(), N))
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    else:
        _check_out(out, (M, N), torch.float32, x.device, "out", rows_may_be_strided=True)
    L = _lib.lib()
    nbytes = 0 if x.single_segment_rows else L.reco_csr_linear_workspace_bytes(x.n_seg, N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    with torch.cuda.device(x.device):
        check(L.reco_csr_linear_f32(_p(x.row_ptr), _p(x.col), _p(x.val), M, K, _p(x.row_seg_ptr), _p(x.seg_row), _p(x.seg_begin),
                                    _p(x.seg_end), x.n_seg, _p(x.order), 0 if x.order is None else x.order.numel(),
                                    1 if x.single_segment_rows else 0, _p(weight_t), N, _p(bias), N, ACTS[act], lo, hi,
                                    _p(out), out.stride(0) if M > 1 else N, _p(ws), nbytes, _stream(x.device)))
    return out


def autorec_forward(x, we, be, wd, bd, clamp=None, precision=None):
    """decoder(sigmoid(encoder(x))) (+clamp) -- src/autorec/utils/model.py:31-33, inference_engine.py:232."""
    x = _f32(x, "input", 2)
    we = _f32(we, "encoder.0.weight", 2); be = _f32(be, "encoder.0.bias", 1)
    wd = _f32(wd, "decoder.0.weight", 2); bd = _f32(bd, "decoder.0.bias", 1)
    B, n_in = x.shape
    hidden = we.shape[0]
    if we.shape[1] != n_in or wd.shape != (n_in, hidden):
        raise ValueError("AutoRec shape mismatch: input %s encoder %s decoder %s" % (tuple(x.shape), tuple(we.shape), tuple(wd.shape)))
    h = torch.empty((B, hidden), dtype=torch.float32, device=x.device)
    out = torch.empty((B, n_in), dtype=torch.float32, device=x.device)
    act, lo, hi = (ACT_NONE, 0.0, 0.0) if clamp is None else (ACT_CLAMP, float(clamp[0]), float(clamp[1]))
    L = _lib.lib()
    nbytes = L.reco_autorec_workspace_bytes(B, n_in, hidden)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    with torch.cuda.device(x.device):
        check(L.reco_autorec_forward(_p(x), B, _p(we), _p(be), _p(wd), _p(bd), n_in, hidden, act, lo, hi,
                                     4 if PRECISIONS[precision or tower_precision()] else 0,
                                     _p(h), _p(out), _p(ws), nbytes, _stream(x.device)))
    return out


# ----------------------------------------------------------------------------------------------
# K-d top-k
# ----------------------------------------------------------------------------------------------
def topk_rows(scores, k, exclude=None):
    """(ids int32 [rows,k], vals [rows,k]): descending score, ascending position among equals.

    exclude (bool/uint8 [rows,N]) applies helper.py:313 (score*(~seen) - seen*1e10) before the select.
    """
    if isinstance(scores, torch.Tensor) and scores.dim() == 2 and scores.is_cuda and scores.dtype == torch.float32 \
            and scores.stride(1) == 1 and scores.stride(0) >= scores.shape[1]:
        pass                                        # a column block of a wider row-major buffer is read in place: ld = stride(0)
    else:
        scores = _f32(scores, "scores", 2)
    rows, N = scores.shape
    ld = scores.stride(0) if rows > 1 else N
    if exclude is not None:
        exclude = _mask(exclude, scores)
        if tuple(exclude.shape) != (rows, N):
            raise ValueError("exclude mask must have shape (%d, %d), got %s" % (rows, N, tuple(exclude.shape)))
    ids = torch.empty((rows, k), dtype=torch.int32, device=scores.device)
    vals = torch.empty((rows, k), dtype=torch.float32, device=scores.device)
    L = _lib.lib()
    nbytes = L.reco_topk_workspace_bytes(rows, N, k)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=scores.device)
    with torch.cuda.device(scores.device):
        check(L.reco_topk_rows(_p(scores), rows, N, ld, k, _p(exclude), N, _p(ids), _p(vals), _p(ws), nbytes,
                               _stream(scores.device)))
    return ids, vals


def topk_rows_seen(scores, k, seen=None, rows=None, check_ids=True):
    """Per-user top-k of a score matrix with the ranking evaluation's seen-item mask taken straight from the float rating
    matrix (helper.py:272,313: seen = train_mat > 0; score*(~seen) - seen*1e10) and an optional user selection.

    scores [U, N] (row-strided views are read in place); seen: float32 [U, N] rating matrix or None; rows: int64 user ids or
    None (all users in order).  Returns (ids int32 [R,k], vals [R,k]); no mask tensor and no gathered copies are built.
    """
    if not (isinstance(scores, torch.Tensor) and scores.dim() == 2 and scores.is_cuda and scores.dtype == torch.float32
            and scores.stride(1) == 1 and scores.stride(0) >= scores.shape[1]):
        scores = _f32(scores, "scores", 2)
    U, N = scores.shape
    ld = scores.stride(0) if U > 1 else N
    ld_seen = 0
    if seen is not None:
        if not (isinstance(seen, torch.Tensor) and seen.dim() == 2 and seen.is_cuda and seen.dtype == torch.float32
                and seen.stride(1) == 1 and seen.stride(0) >= seen.shape[1]):
            seen = _f32(seen, "seen", 2)
        if tuple(seen.shape) != (U, N) or seen.device != scores.device:
            raise ValueError("seen must be a float32 [%d, %d] matrix on %s, got %s" % (U, N, scores.device, tuple(seen.shape)))
        ld_seen = seen.stride(0) if U > 1 else N
    if rows is not None:
        rows = _i64(rows, "rows", scores.device)
    R = U if rows is None else rows.numel()
    ids = torch.empty((R, k), dtype=torch.int32, device=scores.device)
    vals = torch.empty((R, k), dtype=torch.float32, device=scores.device)
    if R == 0:                  # an empty selection (its data pointer is NULL, which the C ABI reads as "all rows")
        return ids, vals
    L = _lib.lib()
    nbytes = L.reco_topk_workspace_bytes(R, N, k)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=scores.device)
    oob = torch.zeros(1, dtype=torch.int32, device=scores.device) if (check_ids and rows is not None) else None
    with torch.cuda.device(scores.device):
        check(L.reco_topk_rows_seen(_p(scores), U, N, ld, k, _p(rows), R, _p(seen), ld_seen, _p(ids), _p(vals), _p(oob),
                                    _p(ws), nbytes, _stream(scores.device)))
    if oob is not None and int(oob.item()):
        raise IndexError("index out of range in self")
    return ids, vals


def _check_out(t, shape, dtype, device, name, rows_may_be_strided=False):
    """A caller-provided output tensor is handed to the kernels as a raw pointer: its extents must be what they assume."""
    ok = isinstance(t, torch.Tensor) and tuple(t.shape) == tuple(shape) and t.dtype == dtype and t.device == device
    if ok and t.numel():
        if rows_may_be_strided:
            ok = (shape[1] == 1 or t.stride(1) == 1) and (shape[0] == 1 or t.stride(0) >= shape[1])
        else:
            ok = t.is_contiguous()
    if not ok:
        raise ValueError("%s must be a %s%s tensor of shape %s on %s" % (
            name, "" if rows_may_be_strided else "contiguous ", str(dtype).replace("torch.", ""), tuple(shape), device))


def _mask(m, like):
    if not isinstance(m, torch.Tensor):
        m = torch.as_tensor(m)
    if m.dtype not in (torch.bool, torch.uint8):
        raise TypeError("exclude mask must be bool or uint8")
    m = m.to(like.device)
    if m.dim() != 2:
        raise ValueError("exclude mask must be 2-D")
    if not m.is_contiguous():
        m = m.contiguous()
    return m.view(torch.uint8) if m.dtype == torch.bool else m       # bool is one byte holding 0 / 1: no conversion pass


# ----------------------------------------------------------------------------------------------
# K-b tower handle
# ----------------------------------------------------------------------------------------------
PRECISIONS = {"f32": _lib.PREC_F32, "bf16x3": _lib.PREC_SPLIT_BF16X3}
_default_precision = [os.environ.get("RECO_TOWER_PRECISION", "f32")]


def set_tower_precision(mode):
    """How the MFMA layers of the fused tower multiply (include/reco_hip.h, RECO_PREC_*): "f32" (exact fp32 MFMA, the
    default) or "bf16x3" (fp32 operands split exactly into three bf16 pieces, six cross terms on the bf16 MFMA, fp32
    accumulate; same parity gates, ~1.7x the throughput).  Applies to towers built afterwards; returns the old mode."""
    if mode not in PRECISIONS:
        raise ValueError("precision must be one of %s, got %r" % (sorted(PRECISIONS), mode))
    old = _default_precision[0]
    _default_precision[0] = mode
    return old


def tower_precision():
    mode = _default_precision[0]
    if mode not in PRECISIONS:
        raise ValueError("RECO_TOWER_PRECISION must be one of %s, got %r" % (sorted(PRECISIONS), mode))
    return mode


class Tower:
    """Owns a reco_tower_t.  Keeps the weight tensors alive (the handle borrows their memory)."""

    def __init__(self, *, user_gmf=None, item_gmf=None, user_mlp=None, item_mlp=None, layers=(), w_predict, b_predict,
                 dg=None, dm=None, act=None, lo=1.0, hi=5.0, precision=None):
        require_gpu()
        self.precision = precision or tower_precision()
        if self.precision not in PRECISIONS:
            raise ValueError("precision must be one of %s, got %r" % (sorted(PRECISIONS), self.precision))
        dev = w_predict.device
        self.device = dev
        keep = []

        def f(t, name, dim):
            if t is None:
                return None
            t = _f32(t, name, dim)
            keep.append(t)
            return t

        user_gmf = f(user_gmf, "user_gmf", 2); item_gmf = f(item_gmf, "item_gmf", 2)
        user_mlp = f(user_mlp, "user_mlp", 2); item_mlp = f(item_mlp, "item_mlp", 2)
        d = TowerDesc()
        d.user_gmf = user_gmf.data_ptr() if user_gmf is not None else None
        d.item_gmf = item_gmf.data_ptr() if item_gmf is not None else None
        d.user_mlp = user_mlp.data_ptr() if user_mlp is not None else None
        d.item_mlp = item_mlp.data_ptr() if item_mlp is not None else None
        tab_u = user_gmf if user_gmf is not None else user_mlp
        tab_i = item_gmf if item_gmf is not None else item_mlp
        d.n_users = tab_u.shape[0] if tab_u is not None else 0
        d.n_items = tab_i.shape[0] if tab_i is not None else 0
        d.dg = int(dg if dg is not None else (user_gmf.shape[1] if user_gmf is not None else 0))
        d.dm = int(dm if dm is not None else (user_mlp.shape[1] if user_mlp is not None else 0))
        if len(layers) > 3:
            raise ValueError("reco_hip: at most 3 MLP layers are supported, got %d" % len(layers))
        # the C side sees raw pointers only: every extent it will assume is checked here
        for name, t, rows, width in (("user_gmf", user_gmf, d.n_users, d.dg), ("item_gmf", item_gmf, d.n_items, d.dg),
                                     ("user_mlp", user_mlp, d.n_users, d.dm), ("item_mlp", item_mlp, d.n_items, d.dm)):
            if t is not None and tuple(t.shape) != (rows, width):
                raise ValueError("%s has shape %s, expected (%d, %d)" % (name, tuple(t.shape), rows, width))
        for t in keep:
            if t.device != dev:
                raise ValueError("all tower tensors must live on %s, got one on %s" % (dev, t.device))
        d.n_layers = len(layers)
        fan_in = 2 * d.dm
        for i, (w, b) in enumerate(layers):
            w = f(w, "W[%d]" % i, 2); b = f(b, "b[%d]" % i, 1)
            if w.shape[1] != fan_in or b.numel() != w.shape[0] or w.device != dev or b.device != dev:
                raise ValueError("MLP layer %d: weight %s / bias %s do not fit an input of width %d on %s"
                                 % (i, tuple(w.shape), tuple(b.shape), fan_in, dev))
            d.W[i] = w.data_ptr(); d.b[i] = b.data_ptr(); d.layer_out[i] = w.shape[0]
            fan_in = w.shape[0]
        if w_predict.numel() != d.dg + (fan_in if layers else 0) or b_predict.numel() != 1:
            raise ValueError("predict layer has %d weights / %d biases, expected %d / 1 (gmf %d + mlp %d)"
                             % (w_predict.numel(), b_predict.numel(), d.dg + (fan_in if layers else 0), d.dg, fan_in if layers else 0))
        wp = f(w_predict.reshape(-1), "w_predict", 1); bp = f(b_predict.reshape(-1), "b_predict", 1)
        d.w_predict = wp.data_ptr(); d.b_predict = bp.data_ptr()
        d.act = ACTS[act]; d.lo = lo; d.hi = hi
        d.precision = PRECISIONS[self.precision]
        self.n_users, self.n_items = int(d.n_users), int(d.n_items)
        self.dg, self.dm = int(d.dg), int(d.dm)
        self._keep = keep
        self._h = ctypes.c_void_p()
        self._L = _lib.lib()
        with torch.cuda.device(dev):
            check(self._L.reco_tower_create(ctypes.byref(d), _stream(dev), ctypes.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._L.reco_tower_destroy(h)
            self._h = None

    def set_reserved_cus(self, cus):
        """Leave `cus` CUs without a tile-kernel workgroup (reco_tower_set_reserved_cus): room for RCCL's all-gather kernel while a
        persistent tile launch is in flight.  0 = use every CU (the default)."""
        check(self._L.reco_tower_set_reserved_cus(self._h, int(cus)))
        self.reserved_cus = int(cus)

    def scale_is_stale(self):
        """True when the MLP tables / first-layer weights were changed in place since the handle was built in a way that its
        layer-1 scale no longer covers (reco_tower_check_scale: one pass over the tables, one device-to-host sync)."""
        flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(self._L.reco_tower_check_scale(self._h, _p(flag), _stream(self.device)))
        return bool(int(flag.item()))

    def _oob(self, check_ids):
        """Per-call counter of clamped ids (None: the caller validated its ids, nothing is counted and nothing synchronises)."""
        return torch.zeros(1, dtype=torch.int32, device=self.device) if check_ids else None

    @staticmethod
    def _raise_if_oob(oob):
        if oob is not None and int(oob.item()):
            raise IndexError("index out of range in self")

    def score_pairs(self, user, item, check_ids=True):
        user = _i64(user, "user", self.device); item = _i64(item, "item", self.device)
        if user.numel() != item.numel():
            raise ValueError("user and item must have the same length (%d vs %d)" % (user.numel(), item.numel()))
        out = torch.empty(user.numel(), dtype=torch.float32, device=self.device)
        oob = self._oob(check_ids)
        with torch.cuda.device(self.device):
            check(self._L.reco_tower_score_pairs(self._h, _p(user), _p(item), user.numel(), _p(out), _p(oob), _stream(self.device)))
            self._raise_if_oob(oob)
        return out

    def score_rows(self, u_gmf, i_gmf, u_mlp, i_mlp):
        ref = u_mlp if u_mlp is not None else u_gmf
        B = ref.shape[0]
        tens = [None if t is None else _f32(t, "rows", 2) for t in (u_gmf, i_gmf, u_mlp, i_mlp)]
        for name, t, width in (("u_gmf", tens[0], self.dg), ("i_gmf", tens[1], self.dg), ("u_mlp", tens[2], self.dm), ("i_mlp", tens[3], self.dm)):
            if width and (t is None or tuple(t.shape) != (B, width) or t.device != self.device):
                raise ValueError("%s must be a [%d, %d] float32 tensor on %s, got %s"
                                 % (name, B, width, self.device, None if t is None else tuple(t.shape)))
        out = torch.empty(B, dtype=torch.float32, device=self.device)
        same = (tens[0] is not None and tens[2] is not None and tens[0].data_ptr() == tens[2].data_ptr()
                and tens[1].data_ptr() == tens[3].data_ptr())            # the Hybrid forward: GMF and MLP rows are the same latents
        nbytes = self._L.reco_tower_rows_workspace_bytes_for(self._h, 1 if same else 0, B)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device) if nbytes else None
        with torch.cuda.device(self.device):
            check(self._L.reco_tower_score_rows(self._h, _p(tens[0]), _p(tens[1]), _p(tens[2]), _p(tens[3]), B, _p(out),
                                                _p(ws), nbytes, _stream(self.device)))
        return out

    def _row_blocks(self, u_gmf, i_gmf, u_mlp, i_mlp):
        U = (u_mlp if u_mlp is not None else u_gmf).shape[0]
        N = (i_mlp if i_mlp is not None else i_gmf).shape[0]
        tens = [None if t is None else _f32(t, "rows", 2) for t in (u_gmf, i_gmf, u_mlp, i_mlp)]
        for name, t, rows, width in (("u_gmf", tens[0], U, self.dg), ("i_gmf", tens[1], N, self.dg),
                                     ("u_mlp", tens[2], U, self.dm), ("i_mlp", tens[3], N, self.dm)):
            if width and (t is None or tuple(t.shape) != (rows, width) or t.device != self.device):
                raise ValueError("%s must be a [%d, %d] float32 tensor on %s, got %s"
                                 % (name, rows, width, self.device, None if t is None else tuple(t.shape)))
        return tens, U, N

    def score_matrix_rows(self, u_gmf, i_gmf, u_mlp, i_mlp, out=None):
        """scores[u, n] for every (row u of the user blocks, row n of the item blocks): score_matrix with per-call tables."""
        tens, U, N = self._row_blocks(u_gmf, i_gmf, u_mlp, i_mlp)
        if out is None:
            out = torch.empty((U, N), dtype=torch.float32, device=self.device)
        else:
            _check_out(out, (U, N), torch.float32, self.device, "out", rows_may_be_strided=True)
        nbytes = self._L.reco_tower_matrix_workspace_bytes(U, N)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(self._L.reco_tower_score_matrix_rows(self._h, _p(tens[0]), _p(tens[1]), _p(tens[2]), _p(tens[3]), U, N, _p(out),
                                                       out.stride(0) if U > 1 else N, _p(ws), nbytes, _stream(self.device)))
        return out

    def score_topk_rows(self, u_gmf, i_gmf, u_mlp, i_mlp, k=10, exclude=None, ws=None, seen=None):
        """(ids int32 [U,k] = row numbers of the item blocks, scores [U,k]): score_topk with per-call tables.
        seen: the float32 [U, N] rating matrix itself (entry > 0 = seen item, helper.py:272,313) instead of a byte mask."""
        tens, U, N = self._row_blocks(u_gmf, i_gmf, u_mlp, i_mlp)
        if seen is not None:
            if exclude is not None:
                raise ValueError("give the seen items as a mask (exclude) or as the rating matrix (seen), not both")
            if not (isinstance(seen, torch.Tensor) and seen.dim() == 2 and seen.is_cuda and seen.dtype == torch.float32
                    and seen.stride(1) == 1 and seen.stride(0) >= seen.shape[1]):
                seen = _f32(seen, "seen", 2)
            if tuple(seen.shape) != (U, N) or seen.device != self.device:
                raise ValueError("seen must be a float32 [%d, %d] matrix on %s, got %s" % (U, N, self.device, tuple(seen.shape)))
            ids = torch.empty((U, k), dtype=torch.int32, device=self.device)
            vals = torch.empty((U, k), dtype=torch.float32, device=self.device)
            if ws is None:
                ws = self.topk_workspace(U, N, k)
            with torch.cuda.device(self.device):
                check(self._L.reco_tower_score_topk_rows_seen(self._h, _p(tens[0]), _p(tens[1]), _p(tens[2]), _p(tens[3]), U, N, k, _p(seen),
                                                              seen.stride(0) if U > 1 else N, _p(ids), _p(vals), _p(ws), ws.numel(),
                                                              _stream(self.device)))
            return ids, vals
        if exclude is not None:
            exclude = _mask(exclude, tens[2] if tens[2] is not None else tens[0])
            if tuple(exclude.shape) != (U, N):
                raise ValueError("exclude mask must have shape (%d, %d), got %s" % (U, N, tuple(exclude.shape)))
        ids = torch.empty((U, k), dtype=torch.int32, device=self.device)
        vals = torch.empty((U, k), dtype=torch.float32, device=self.device)
        if ws is None:
            ws = self.topk_workspace(U, N, k)
        with torch.cuda.device(self.device):
            check(self._L.reco_tower_score_topk_rows(self._h, _p(tens[0]), _p(tens[1]), _p(tens[2]), _p(tens[3]), U, N, k,
                                                     _p(exclude), N, _p(ids), _p(vals), _p(ws), ws.numel(), _stream(self.device)))
        return ids, vals

    def _users(self, users):
        """users: None (all), int range tuple (base, count), or id tensor."""
        if users is None:
            return None, 0, self.n_users
        if isinstance(users, tuple):
            return None, int(users[0]), int(users[1])
        users = _i64(users, "users", self.device)
        return users, 0, users.numel()

    def score_matrix(self, users=None, cands=None, check_ids=True, out=None):
        uptr, base, U = self._users(users)
        if cands is not None:
            cands = _i64(cands, "candidates", self.device)
        N = cands.numel() if cands is not None else self.n_items
        if out is None:
            out = torch.empty((U, N), dtype=torch.float32, device=self.device)
        else:
            _check_out(out, (U, N), torch.float32, self.device, "out", rows_may_be_strided=True)
        nbytes = self._L.reco_tower_matrix_workspace_bytes(U, N)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=self.device)
        oob = self._oob(check_ids)
        with torch.cuda.device(self.device):
            check(self._L.reco_tower_score_matrix(self._h, _p(uptr), base, U, _p(cands), N, _p(out), out.stride(0),
                                                  _p(oob), _p(ws), nbytes, _stream(self.device)))
            self._raise_if_oob(oob)
        return out

    def topk_workspace(self, U, N, k, fused=False):
        if fused:
            nbytes = self._L.reco_tower_topk_fused_workspace_bytes(self._h, U, N, k)
        else:
            nbytes = self._L.reco_tower_topk_workspace_bytes(U, N, k)
        return torch.empty(max(nbytes, 1), dtype=torch.uint8, device=self.device)

    def score_topk(self, users=None, k=10, cands=None, exclude=None, check_ids=True, ws=None, out=None, fused=False):
        """(ids int32 [U,k] = positions in `cands` (item ids when cands is None), scores [U,k]).

        fused=True: the one-kernel form that keeps no score rows in memory (reco_tower_score_topk_fused: same results, far
        smaller workspace and HBM traffic, 1-3 % more time); mask-free calls on the exact MFMA tower only."""
        if fused and exclude is not None:
            raise ValueError("the fused score + top-k form takes no exclude mask")
        uptr, base, U = self._users(users)
        if cands is not None:
            cands = _i64(cands, "candidates", self.device)
        N = cands.numel() if cands is not None else self.n_items
        if exclude is not None:
            exclude = _mask(exclude, self._keep[0])
            if tuple(exclude.shape) != (U, N):
                raise ValueError("exclude mask must have shape (%d, %d), got %s" % (U, N, tuple(exclude.shape)))
        if out is None:
            ids = torch.empty((U, k), dtype=torch.int32, device=self.device)
            vals = torch.empty((U, k), dtype=torch.float32, device=self.device)
        else:
            ids, vals = out
            _check_out(ids, (U, k), torch.int32, self.device, "out[0] (ids)")
            _check_out(vals, (U, k), torch.float32, self.device, "out[1] (scores)")
        if ws is None:
            ws = self.topk_workspace(U, N, k, fused)
        elif not (isinstance(ws, torch.Tensor) and ws.dtype == torch.uint8 and ws.device == self.device and ws.is_contiguous()):
            raise ValueError("ws must be the uint8 tensor topk_workspace() returned, on %s" % self.device)
        oob = self._oob(check_ids)
        with torch.cuda.device(self.device):
            if fused:
                check(self._L.reco_tower_score_topk_fused(self._h, _p(uptr), base, U, _p(cands), N, k, _p(ids), _p(vals),
                                                          _p(oob), _p(ws), ws.numel(), _stream(self.device)))
            else:
                check(self._L.reco_tower_score_topk(self._h, _p(uptr), base, U, _p(cands), N, k, _p(exclude), N, _p(ids),
                                                    _p(vals), _p(oob), _p(ws), ws.numel(), _stream(self.device)))
            self._raise_if_oob(oob)
        return ids, vals


class GenericTower:
    """Same interface as Tower for NCF shapes the fused MFMA kernel does not cover (factor_num != 32 or
    num_layers != 3, e.g. the 8/16/64/128 embedding sweep of src/ncf/ncf-embedding.ipynb).

    Still all-HIP, but unfused: K-a gather -> K-c Linear+ReLU per layer -> K-c predict (N = 1).  The GMF product is
    written straight into the predict layer's concat buffer, the last MLP layer writes the other column block.
    """

    PAIRS_PER_CHUNK = 1 << 21

    def __init__(self, *, user_gmf=None, item_gmf=None, user_mlp=None, item_mlp=None, layers=(), w_predict, b_predict,
                 act=None, lo=1.0, hi=5.0):
        require_gpu()
        self.device = w_predict.device
        self.ug = None if user_gmf is None else _f32(user_gmf, "user_gmf", 2)
        self.ig = None if item_gmf is None else _f32(item_gmf, "item_gmf", 2)
        self.um = None if user_mlp is None else _f32(user_mlp, "user_mlp", 2)
        self.im = None if item_mlp is None else _f32(item_mlp, "item_mlp", 2)
        self.layers = [(_f32(w, "W", 2), _f32(b, "b", 1)) for w, b in layers]
        self.wp = _f32(w_predict.reshape(1, -1), "w_predict", 2)
        self.bp = _f32(b_predict.reshape(-1), "b_predict", 1)
        self.act, self.lo, self.hi = act, lo, hi
        tab_u = self.ug if self.ug is not None else self.um
        tab_i = self.ig if self.ig is not None else self.im
        self.n_users, self.n_items = tab_u.shape[0], tab_i.shape[0]
        self.dg = 0 if self.ug is None else self.ug.shape[1]
        self.dout = self.layers[-1][0].shape[0] if self.layers else 0
        if self.dg + self.dout != self.wp.shape[1]:
            raise ValueError("predict layer width %d does not match gmf %d + mlp %d" % (self.wp.shape[1], self.dg, self.dout))

    def score_pairs(self, user, item, check_ids=True):
        user = _i64(user, "user", self.device); item = _i64(item, "item", self.device)
        if user.numel() != item.numel():
            raise ValueError("user and item must have the same length (%d vs %d)" % (user.numel(), item.numel()))
        B = user.numel()
        out = torch.empty(B, dtype=torch.float32, device=self.device)
        for s in range(0, B, self.PAIRS_PER_CHUNK):
            u = user[s:s + self.PAIRS_PER_CHUNK]; i = item[s:s + self.PAIRS_PER_CHUNK]
            n = u.numel()
            width = self.dg + self.dout
            pad = (-width) % 4                                   # keep every row of the concat buffer 16-byte aligned
            concat = torch.empty((n, width + pad), dtype=torch.float32, device=self.device)[:, :width]
            gmf_view = concat[:, :self.dg] if self.dg else None
            _, x = neumf_gather(self.ug, self.ig, self.um, self.im, u, i, check_ids, gmf_out=gmf_view)
            for li, (w, b) in enumerate(self.layers):
                last = li + 1 == len(self.layers)
                x = linear(x, w, b, "relu", out=concat[:, self.dg:] if last else None)
            linear(concat, self.wp, self.bp, self.act, self.lo, self.hi, out=out[s:s + n].view(n, 1))
        return out

    def _users(self, users):
        if users is None:
            return torch.arange(self.n_users, device=self.device)
        if isinstance(users, tuple):
            return torch.arange(users[0], users[0] + users[1], device=self.device)
        return _i64(users, "users", self.device)

    def score_matrix(self, users=None, cands=None, check_ids=True, out=None):
        users = self._users(users)
        cands = torch.arange(self.n_items, device=self.device) if cands is None else _i64(cands, "candidates", self.device)
        U, N = users.numel(), cands.numel()
        if out is None:
            out = torch.empty((U, N), dtype=torch.float32, device=self.device)
        step = max(1, self.PAIRS_PER_CHUNK // max(N, 1))
        for s in range(0, U, step):
            u = users[s:s + step]
            out[s:s + step] = self.score_pairs(u.repeat_interleave(N), cands.repeat(u.numel()), check_ids).view(u.numel(), N)
        return out

    def topk_workspace(self, U, N, k):
        return None

    def score_topk(self, users=None, k=10, cands=None, exclude=None, check_ids=True, ws=None, out=None):
        users = self._users(users)
        U = users.numel()
        N = self.n_items if cands is None else len(cands)
        ids = torch.empty((U, k), dtype=torch.int32, device=self.device) if out is None else out[0]
        vals = torch.empty((U, k), dtype=torch.float32, device=self.device) if out is None else out[1]
        step = max(1, (1 << 26) // max(N, 1))
        for s in range(0, U, step):
            sc = self.score_matrix(users[s:s + step], cands, check_ids)
            i, v = topk_rows(sc, k, None if exclude is None else exclude[s:s + step])
            ids[s:s + step] = i; vals[s:s + step] = v
        return ids, vals
