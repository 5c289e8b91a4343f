"""Layer / model API mirror of dev/py/complex.py + model.py.

CPU part: TF-style variable naming, shapes, error behaviour (no compute).
GPU part (-m gpu): every layer against the oracle's literal restatement; the composable model against
the fused engine."""
import types

import numpy as np
import pytest
import torch

from oracle import dccn_oracle as O


def flags(**kw):
    base = dict(nsymbol=7, nfft=64, longcp=True, pilot="lte", npilot=8, nguard=8, nbits=2, channel="EPA", cp=True,
                nfilter=64)
    base.update(kw)
    return types.SimpleNamespace(**base)


def relerr(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.abs(got - ref).max()) / max(float(np.abs(ref).max()), 1e-30)


# ---- CPU ---------------------------------------------------------------------------------------
def test_variable_names_match_reference_checkpoint_names():
    from dl_ofdm_amd import ofdm
    from dl_ofdm_amd.engine import PARAM_NAMES
    from dl_ofdm_amd.model import OfdmDenseRx
    for nbits, cp in ((1, True), (2, True), (4, False)):
        F = flags(nbits=nbits, cp=cp)
        m = OfdmDenseRx(F, ofdm.ofdm_tx(F), device="cpu")
        assert tuple(m.store.names()) == PARAM_NAMES
        kin = 80 if cp else 64
        assert m.store.meta["fft_like/conv3d/kernel"]["tf_shape"] == (1, kin, 1, kin, 128)
        assert m.store.meta["fft_like/conv3d/kernel"]["live_taps"] == ((0,), ((kin - 1) // 2,))
        assert m.store.meta["demodulation/dense/kernel"]["shape"] == (896, 640)
        assert m.store.meta["demodulation/dense_1/kernel"]["shape"] == (2 ** nbits + 2, 2 * nbits)
        lim = np.sqrt(6.0 / (kin * kin + kin * 128))               # glorot over the FULL conv3d kernel (A.7)
        w = m.store.tensor("fft_like/conv3d/kernel").detach().numpy()
        assert np.abs(w).max() <= lim and np.abs(w).max() > 0.9 * lim
        assert float(m.store.tensor("demodulation/dense/bias").abs().max()) == 0.0


def test_scope_auto_naming_like_tf_layers():
    from dl_ofdm_amd.complex import VariableStore
    st = VariableStore(device="cpu")
    with st.scope("Equalizer"):
        assert [st.layer_name("dense"), st.layer_name("conv3d"), st.layer_name("dense"), st.layer_name("conv3d")] == \
            ["Equalizer/dense", "Equalizer/conv3d", "Equalizer/dense_1", "Equalizer/conv3d_1"]
    st.begin()
    with st.scope("Equalizer"):
        assert st.layer_name("dense") == "Equalizer/dense"


def test_layer_argument_errors():
    from dl_ofdm_amd.complex import VariableStore, layers_conv1d_complex, layers_conv2d_complex
    st = VariableStore(device="cpu")
    with pytest.raises(TypeError):
        layers_conv2d_complex(torch.zeros(2, 7, 1, 80), 4, (1, 80), scope=st)        # real rank-4: not complex64
    with pytest.raises(NameError):
        layers_conv2d_complex(torch.zeros(2, 7, 1, 80, 2), 4, [1, 80], scope=st)     # kernel must be int or tuple
    with pytest.raises(NameError):
        layers_conv1d_complex(torch.zeros(2, 7, 80, 3), 4, 3, scope=st)
    with pytest.raises(AssertionError):
        layers_conv1d_complex(torch.zeros(2, 7, 80, 2), 4, (3,), scope=st)


def test_live_taps():
    from dl_ofdm_amd.complex import _live_taps, tf_padding
    out, p0, _ = tf_padding(1, 80, 1, "same")
    assert (out, p0) == (1, 39) and _live_taps(1, 80, 1, out, p0) == [39]
    out, p0, _ = tf_padding(1, 64, 1, "same")
    assert _live_taps(1, 64, 1, out, p0) == [31]
    out, p0, _ = tf_padding(7, 7, 1, "same")
    assert _live_taps(7, 7, 1, out, p0) == list(range(7))
    out, p0, _ = tf_padding(64, 64, 1, "valid")
    assert out == 1 and _live_taps(64, 64, 1, out, p0) == list(range(64))


# ---- GPU ---------------------------------------------------------------------------------------
def _full_kernel(store, name, rng):
    """Embed the stored live taps into a TF-shaped kernel whose dead taps hold random values."""
    meta = store.meta[name + "/kernel"]
    tl, tw = meta["live_taps"]
    full = rng.randn(*meta["tf_shape"])
    live = store.tensor(name + "/kernel").detach().cpu().numpy().astype(np.float64)
    for ia, a in enumerate(tl):
        for ib, b in enumerate(tw):
            full[a, b, 0] = live[ia, ib]
    return full


@pytest.mark.gpu
@pytest.mark.parametrize("shape,F,kernal,strides,padding", [
    ((3, 7, 1, 80, 2), 64, (1, 80), 1, "same"),         # receiver fft_like
    ((3, 7, 64, 1, 2), 64, (1, 64), 1, "valid"),        # equalizer frequency-domain conversion
    ((2, 7, 64, 1, 2), 1, (7, 64), (1, 1), "same"),     # equalizer 2-D smoothing, F = 1
    ((2, 9, 10, 3, 2), 5, (3, 2), (2, 1), "valid"),
    ((2, 9, 10, 3, 2), 5, 3, 1, "same"),
])
def test_layers_conv2d_complex_vs_literal(shape, F, kernal, strides, padding):
    from dl_ofdm_amd.complex import VariableStore, layers_conv2d_complex
    rng = np.random.RandomState(0)
    x = rng.randn(*shape).astype(np.float32)
    st = VariableStore(seed=3)
    xt = torch.as_tensor(x).cuda().requires_grad_()
    y = layers_conv2d_complex(xt, F, kernal, strides=strides, padding=padding, scope=st)
    st.set("conv3d/bias", rng.randn(2 * F))
    st.begin()
    y = layers_conv2d_complex(xt, F, kernal, strides=strides, padding=padding, scope=st)
    full = _full_kernel(st, "conv3d", rng)
    stt = (strides, strides) if isinstance(strides, int) else strides
    ref = O.layers_conv2d_complex_literal(x.astype(np.float64), full, st.tensor("conv3d/bias").detach().cpu().numpy().astype(np.float64), stt, padding)
    assert relerr(y.detach().cpu().numpy(), ref) <= 1e-5
    y.sum().backward()                                                      # gradients reach inputs and variables
    assert xt.grad is not None and st.tensor("conv3d/kernel").grad is not None
    # complex64 input -> complex64 output
    xc = torch.complex(xt.detach()[..., 0], xt.detach()[..., 1])
    st.begin()
    yc = layers_conv2d_complex(xc, F, kernal, strides=strides, padding=padding, scope=st)
    assert yc.dtype == torch.complex64 and torch.equal(torch.view_as_real(yc), y.detach())


@pytest.mark.gpu
def test_conv1d_layers_vs_literal():
    from dl_ofdm_amd.complex import VariableStore, layers_conv1d_complex, nn_conv1d_complex
    rng = np.random.RandomState(1)
    x = rng.randn(4, 11, 6, 2).astype(np.float32)
    st = VariableStore(seed=5)
    y = layers_conv1d_complex(torch.as_tensor(x).cuda(), 7, 3, strides=1, padding="same", scope=st)
    k = st.tensor("conv2d/kernel").detach().cpu().numpy().astype(np.float64).reshape(3, 1, 6, 14)
    b = st.tensor("conv2d/bias").detach().cpu().numpy().astype(np.float64)
    assert relerr(y.detach().cpu().numpy(), O.layers_conv1d_complex_literal(x.astype(np.float64), k, b, 1, "same")) <= 1e-5
    f = rng.randn(3, 6, 1, 2).astype(np.float32)
    z = nn_conv1d_complex(torch.as_tensor(x).cuda(), torch.as_tensor(f).cuda())
    assert relerr(z.detach().cpu().numpy(), O.nn_conv1d_complex(x.astype(np.float64), f.astype(np.float64))) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("shape,F,k,stride,padding", [
    ((4, 37, 6, 2), 8, 5, 1, "same"),          # k = 5: two taps of padding at either end
    ((3, 64, 4, 2), 6, 5, 2, "same"),          # strided (input gradient: col2im route)
    ((2, 33, 8, 2), 4, 7, 1, "valid"),
    ((5, 9, 2, 2), 2, 3, 3, "same"),           # stride = kernel size
    ((2, 130, 64, 2), 64, 5, 1, "same"),       # several 64-deep k-tiles (K = 640), more than one row tile
    ((16, 512, 8, 2), 16, 5, 1, "same"),       # 8192 output positions: many split-K slabs in the weight gradient
    ((3, 20, 3, 2), 5, 3, 1, "same"),          # odd C and F: the im2col operators throughout
])
def test_conv1d_implicit_gemm_vs_literal_and_im2col(shape, F, k, stride, padding):
    """general-k layers_conv1d_complex (complex.py:51-92) as implicit GEMMs -- forward (dccn_cconv_patch_fwd), weight gradient
    (dccn_cconv_patch_bwd_w) and, at stride 1, input gradient (dccn_cconv_patch_bwd_x): output against the literal fp64
    restatement, gradients against the autograd of its torch twin, and the im2col + GEMM + col2im route beside them."""
    from dl_ofdm_amd import complex as CX, ops
    from oracle.torch_ref import layers_conv1d_complex_literal_t
    rng = np.random.RandomState(2)
    x = rng.randn(*shape).astype(np.float32)
    C = shape[2]
    res = {}
    for implicit in (True, False):
        CX.IMPLICIT_GEMM = implicit
        ops._PATCH_BWD_DX_ALWAYS = implicit          # the implicit input gradient wherever it qualifies, not only where it pays
        try:
            st = CX.VariableStore(seed=9)
            xt = torch.as_tensor(x).cuda().requires_grad_()
            CX.layers_conv1d_complex(xt, F, k, strides=stride, padding=padding, scope=st)      # creates the variables
            st.set("conv2d/bias", np.random.RandomState(4).randn(2 * F))
            st.begin()
            y = CX.layers_conv1d_complex(xt, F, k, strides=stride, padding=padding, scope=st)
            g = torch.as_tensor(np.random.RandomState(5).randn(*y.shape).astype(np.float32)).cuda()
            y.backward(g)
            res[implicit] = (y.detach().cpu().numpy(), xt.grad.cpu().numpy(), st.tensor("conv2d/kernel").grad.cpu().numpy(),
                             st.tensor("conv2d/bias").grad.cpu().numpy())
            if implicit:
                assert st.meta["conv2d/kernel"]["live_taps"][0] == tuple(range(k))
                kern = st.tensor("conv2d/kernel").detach().cpu().numpy().astype(np.float64).reshape(k, 1, C, 2 * F)
                bias = st.tensor("conv2d/bias").detach().cpu().numpy().astype(np.float64)
                ref = O.layers_conv1d_complex_literal(x.astype(np.float64), kern, bias, stride, padding)
                assert relerr(res[True][0], ref) <= 1e-5
                x64 = torch.tensor(x.astype(np.float64), requires_grad=True)
                k64, b64 = torch.tensor(kern, requires_grad=True), torch.tensor(bias, requires_grad=True)
                layers_conv1d_complex_literal_t(x64, k64, b64, stride, padding).backward(g.cpu().double())
                grads = (x64.grad.numpy(), k64.grad.numpy().reshape(res[True][2].shape), b64.grad.numpy())
        finally:
            CX.IMPLICIT_GEMM = True
            ops._PATCH_BWD_DX_ALWAYS = False
    assert relerr(res[True][0], res[False][0]) <= 1e-5
    for route in (True, False):
        for got, want in zip(res[route][1:], grads):
            assert relerr(got, want) <= 1e-5, route


@pytest.mark.gpu
def test_conv2d_implicit_gemm_two_tap_axes():
    """taps over both axes, strides (2, 1) / (1, 2) / (1, 1), VALID and SAME: the loaders' (ti, tj) arithmetic in the forward,
    in the weight gradient and (stride 1) in the input gradient's flipped-tap gather; dead taps get no gradient"""
    from dl_ofdm_amd import complex as CX, ops
    from oracle.torch_ref import layers_conv2d_complex_literal_t
    rng = np.random.RandomState(6)
    for shape, F, kern, strides, padding in (((2, 9, 10, 4, 2), 6, (3, 2), (2, 1), "valid"),
                                             ((3, 8, 12, 2, 2), 4, (3, 5), (1, 2), "same"),
                                             ((2, 9, 10, 4, 2), 6, (3, 2), (1, 1), "same"),
                                             ((2, 7, 64, 2, 2), 2, (7, 64), (1, 1), "same"),      # the equaliser's 2-D smoothing shape
                                             ((4, 21, 17, 6, 2), 10, (5, 3), (1, 1), "valid")):
        x = rng.randn(*shape).astype(np.float32)
        st = CX.VariableStore(seed=3)
        xt = torch.as_tensor(x).cuda().requires_grad_()
        assert ops.cconv_patch_supported(xt, 1, 1, kern[0], kern[1], F)
        CX.layers_conv2d_complex(xt, F, kern, strides=strides, padding=padding, scope=st)
        st.set("conv3d/bias", rng.randn(2 * F))
        st.begin()
        y = CX.layers_conv2d_complex(xt, F, kern, strides=strides, padding=padding, scope=st)
        full = _full_kernel(st, "conv3d", rng)
        bias = st.tensor("conv3d/bias").detach().cpu().numpy().astype(np.float64)
        ref = O.layers_conv2d_complex_literal(x.astype(np.float64), full, bias, strides, padding)
        assert relerr(y.detach().cpu().numpy(), ref) <= 1e-5
        g = torch.as_tensor(rng.randn(*y.shape).astype(np.float32)).cuda()
        ops._PATCH_BWD_DX_ALWAYS = True
        try:
            y.backward(g)
        finally:
            ops._PATCH_BWD_DX_ALWAYS = False
        x64 = torch.tensor(x.astype(np.float64), requires_grad=True)
        k64, b64 = torch.tensor(full, requires_grad=True), torch.tensor(bias, requires_grad=True)
        layers_conv2d_complex_literal_t(x64, k64, b64, strides, padding).backward(g.cpu().double())
        tl, tw = st.meta["conv3d/kernel"]["live_taps"]
        kg = k64.grad.numpy()
        live = np.stack([np.stack([kg[a, b, 0] for b in tw]) for a in tl])
        dead = np.abs(kg).sum() - np.abs(live).sum()
        assert dead <= 1e-9 * np.abs(live).sum()                                   # dead taps never meet data
        assert relerr(xt.grad.cpu().numpy(), x64.grad.numpy()) <= 1e-5, (shape, kern, strides)
        assert relerr(st.tensor("conv3d/kernel").grad.cpu().numpy().reshape(live.shape), live) <= 1e-5, (shape, kern, strides)
        assert relerr(st.tensor("conv3d/bias").grad.cpu().numpy(), b64.grad.numpy()) <= 1e-5


@pytest.mark.gpu
def test_conv2d_random_geometries_forward_and_backward():
    """48 seeded random geometries of layers_conv2d_complex (kernel sizes up to and beyond the input extent, strides 1-3, SAME /
    VALID, one or two tap axes, 2-12 channels and filters, outputs of a single position): output and all three gradients of
    the implicit-GEMM route against the fp64 literal / its autograd twin at 1e-5 -- the operand loaders' index arithmetic
    (tap decomposition, padding masks, row decomposition by multiply-shift, flipped taps, ragged tiles) on shapes nobody chose."""
    from dl_ofdm_amd import complex as CX, ops
    from oracle.torch_ref import layers_conv2d_complex_literal_t
    rng = np.random.RandomState(20260928)
    done = 0
    while done < 48:
        B, L, Wd = int(rng.randint(1, 6)), int(rng.randint(1, 24)), int(rng.randint(1, 24))
        C, F = 2 * int(rng.randint(1, 7)), 2 * int(rng.randint(1, 7))
        kern = (int(rng.randint(1, 8)), int(rng.randint(1, 8)))
        strides = (int(rng.randint(1, 4)), int(rng.randint(1, 4))) if rng.rand() < 0.4 else (1, 1)
        padding = "same" if rng.rand() < 0.6 else "valid"
        if padding == "valid" and (kern[0] > L or kern[1] > Wd):
            continue
        x = rng.randn(B, L, Wd, C, 2).astype(np.float32)
        st = CX.VariableStore(seed=int(rng.randint(1, 1000)))
        xt = torch.as_tensor(x).cuda().requires_grad_()
        CX.layers_conv2d_complex(xt, F, kern, strides=strides, padding=padding, scope=st)
        st.set("conv3d/bias", rng.randn(2 * F))
        st.begin()
        ops._PATCH_BWD_DX_ALWAYS = True
        try:
            y = CX.layers_conv2d_complex(xt, F, kern, strides=strides, padding=padding, scope=st)
            g = torch.as_tensor(rng.randn(*y.shape).astype(np.float32)).cuda()
            y.backward(g)
        finally:
            ops._PATCH_BWD_DX_ALWAYS = False
        full = _full_kernel(st, "conv3d", rng)
        bias = st.tensor("conv3d/bias").detach().cpu().numpy().astype(np.float64)
        what = (B, L, Wd, C, F, kern, strides, padding)
        ref = O.layers_conv2d_complex_literal(x.astype(np.float64), full, bias, strides, padding)
        assert relerr(y.detach().cpu().numpy(), ref) <= 1e-5, what
        x64 = torch.tensor(x.astype(np.float64), requires_grad=True)
        k64, b64 = torch.tensor(full, requires_grad=True), torch.tensor(bias, requires_grad=True)
        layers_conv2d_complex_literal_t(x64, k64, b64, strides, padding).backward(g.cpu().double())
        tl, tw = st.meta["conv3d/kernel"]["live_taps"]
        kg = k64.grad.numpy()
        live = np.stack([np.stack([kg[a, b, 0] for b in tw]) for a in tl])
        assert relerr(xt.grad.cpu().numpy(), x64.grad.numpy()) <= 1e-5, what
        assert relerr(st.tensor("conv3d/kernel").grad.cpu().numpy().reshape(live.shape), live) <= 1e-5, what
        assert relerr(st.tensor("conv3d/bias").grad.cpu().numpy(), b64.grad.numpy()) <= 1e-5, what
        done += 1


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,C,F,k,stride,padding", [
    (37, 560, 2, 64, 5, 1, "same"),        # the bench shape's geometry: 59 positions per chunk, 10 chunks per item
    (5, 61, 2, 32, 5, 1, "same"),
    (3, 200, 2, 64, 7, 1, "valid"),
    (4, 333, 2, 64, 5, 2, "same"),         # strided: 121 positions per chunk
    (2, 500, 4, 32, 3, 3, "same"),
    (6, 40, 6, 64, 2, 1, "same"),          # even kernel (asymmetric SAME padding), 24 patch columns
    (1, 1, 2, 64, 3, 1, "same"),           # a single position: two of three taps see padding only
])
def test_conv1d_fused_backward_vs_separate_routes_and_literal(B, L, C, F, k, stride, padding):
    """dccn_cconv1d_bwd (dx, dw, dbias in one pass over dout) against the implicit-GEMM routes (dccn_cconv_patch_bwd_x / _bwd_w)
    and the fp64 autograd twin of the literal TensorFlow formulation at 1e-5."""
    from dl_ofdm_amd import complex as CX, ops
    from oracle.torch_ref import layers_conv1d_complex_literal_t
    rng = np.random.RandomState(B * 1000 + L)
    x = rng.randn(B, L, C, 2).astype(np.float32)
    st = CX.VariableStore(seed=11)
    xt = torch.as_tensor(x).cuda().requires_grad_()
    CX.layers_conv1d_complex(xt, F, k, strides=stride, padding=padding, scope=st)
    st.set("conv2d/bias", rng.randn(2 * F))
    grads = {}
    g = None
    for fused in (True, False):
        ops._PATCH_BWD_FUSED1D = fused
        ops._PATCH_BWD_DX_ALWAYS = True
        try:
            st.begin()
            xt.grad = None
            for n in ("conv2d/kernel", "conv2d/bias"):
                st.tensor(n).grad = None
            y = CX.layers_conv1d_complex(xt, F, k, strides=stride, padding=padding, scope=st)
            if g is None:
                g = torch.as_tensor(rng.randn(*y.shape).astype(np.float32)).cuda()
            y.backward(g)
        finally:
            ops._PATCH_BWD_FUSED1D, ops._PATCH_BWD_DX_ALWAYS = True, False
        grads[fused] = (xt.grad.cpu().numpy().copy(), st.tensor("conv2d/kernel").grad.cpu().numpy().copy(),
                        st.tensor("conv2d/bias").grad.cpu().numpy().copy())
    for a, b in zip(grads[True], grads[False]):
        assert relerr(a, b) <= 1e-5
    tl = list(st.meta["conv2d/kernel"]["live_taps"][0])
    full = rng.randn(k, 1, C, 2 * F)                          # dead taps (they only ever see padding): any value
    full[tl] = st.tensor("conv2d/kernel").detach().cpu().numpy().astype(np.float64).reshape(len(tl), 1, C, 2 * F)
    bias = st.tensor("conv2d/bias").detach().cpu().numpy().astype(np.float64)
    x64 = torch.tensor(x.astype(np.float64), requires_grad=True)
    k64, b64 = torch.tensor(full, requires_grad=True), torch.tensor(bias, requires_grad=True)
    layers_conv1d_complex_literal_t(x64, k64, b64, stride, padding).backward(g.cpu().double())
    assert relerr(grads[True][0], x64.grad.numpy()) <= 1e-5
    assert relerr(grads[True][2], b64.grad.numpy()) <= 1e-5
    live = k64.grad.numpy()[tl]
    assert relerr(grads[True][1].reshape(live.shape), live) <= 1e-5


def test_multiply_shift_division_of_the_patch_loader_is_exact():
    """csrc/gemm_f32_mfma.h patch_div_magic / gemm_kmajor.h patch_div (row -> (b, lo, wo) of the implicit weight-gradient GEMM):
    s = ceil(log2 d), mul = ceil(2^(31+s) / d), q = (n * mul) >> (31 + s) equals n // d for every 0 <= n < 2^31 -- the
    algorithm restated here and checked at the boundaries of every quotient step and on random pairs (the device code itself is
    exercised by the convolution tests above)."""
    def magic(d):
        s = 0
        while (1 << s) < d:
            s += 1
        shift = 31 + s
        mul = ((1 << shift) + d - 1) // d
        assert mul < (1 << 32)
        return mul, shift
    rng = np.random.RandomState(11)
    ds = [1, 2, 3, 5, 7, 64, 73, 560, 4095, 4096, 65535, 65537, (1 << 31) - 1, 1 << 30] + [int(v) for v in rng.randint(1, 1 << 31, 200)]
    top = (1 << 31) - 1
    for d in ds:
        mul, shift = magic(d)
        ns = {0, 1, d - 1, d, d + 1, top, top - 1, (top // d) * d, max((top // d) * d - 1, 0)}
        ns |= {int(v) for v in rng.randint(0, 1 << 31, 50)}
        ns |= {min(int(q) * d + r, top) for q in rng.randint(0, max(top // d, 1) + 1, 20) for r in (0, d - 1)}
        for n in ns:
            assert (n * mul) >> shift == n // d, (n, d)


def test_patch_backward_route_bits():
    """dccn_cconv_patch_bwd_supported: bit 0 weight gradient, bit 1 input gradient as an implicit GEMM (any stride since round
    6), bit 2 that route expected to beat GEMM + col2im"""
    from dl_ofdm_amd import _lib
    f = _lib.load().dccn_cconv_patch_bwd_supported
    assert f(8, 560, 1, 2, 560, 1, 5, 1, 1, 1, 64) == 7          # 2 channels: 16-column tiles (cconv_dx_narrow.h), five taps deep
    assert f(8, 560, 1, 16, 560, 1, 5, 1, 1, 1, 32) == 7
    assert f(8, 560, 1, 64, 560, 1, 5, 1, 1, 1, 64) == 7
    assert f(8, 560, 1, 16, 280, 1, 5, 1, 2, 1, 32) == 7         # strided, 32 columns: narrow tiles, two phases of 2-3 taps
    assert f(8, 560, 1, 64, 280, 1, 5, 1, 2, 1, 32) == 3         # strided, 128 columns: inverse-stride gather works, col2im is cheaper
    assert f(8, 7, 64, 2, 7, 64, 7, 64, 1, 1, 2) == 3            # 448 taps x 4 columns: 16-column tiles stay 3/4 empty 448 times over
    assert f(8, 560, 1, 3, 560, 1, 5, 1, 1, 1, 32) == 0          # odd channel count: no float4 pieces


def test_conv1d_fused_backward_applicability():
    """dccn_cconv1d_bwd_supported (host-side planning, no GPU): few-channel 1-D layers with 2F = 64 or 128 and at most 30 patch
    columns; everything else stays with the separate implicit-GEMM routes"""
    from dl_ofdm_amd import _lib
    lib = _lib.load()
    f = lib.dccn_cconv1d_bwd_supported
    assert f(1170, 560, 2, 560, 5, 1, 64) == 1 and f(8, 333, 2, 167, 5, 2, 64) == 1 and f(4, 40, 6, 40, 2, 1, 32) == 1
    assert f(8, 560, 2, 560, 8, 1, 64) == 0          # 32 patch columns: no room for the row of ones (bias gradient)
    assert f(8, 560, 16, 560, 5, 1, 32) == 0         # 160 patch columns
    assert f(8, 560, 2, 560, 5, 1, 16) == 0 and f(8, 560, 3, 560, 5, 1, 64) == 0      # 2F = 32; odd channel count
    assert f(8, 560, 2, 560, 70, 1, 64) == 0         # 280 patch columns (and more taps than a 64-row chunk owns positions for)
    assert lib.dccn_cconv1d_bwd_workspace_size(64) > lib.dccn_cconv1d_bwd_workspace_size(32) > 0
    assert lib.dccn_cconv1d_bwd(None, None, None, None, None, None, 8, 560, 2, 560, 5, 0, 1, 2, 64, None, 0, None) < 0      # null operands: refused


@pytest.mark.gpu
def test_composable_model_equals_fused_engine():
    from dl_ofdm_amd import ofdm
    from dl_ofdm_amd.engine import RxEngine
    from dl_ofdm_amd.model import OfdmDenseRx
    from dl_ofdm_amd.ops import read_metrics
    F = flags(nbits=2)
    o = ofdm.ofdm_tx(F)
    model = OfdmDenseRx(F, o, seed=4)
    rng = np.random.RandomState(2)
    for n in ("fft_like/conv3d/bias", "demodulation/dense/bias", "demodulation/conv2d/bias"):
        model.store.set(n, rng.uniform(-.05, .05, model.store.tensor(n).shape))
    batch = 200
    x = torch.as_tensor(rng.randn(batch, 7, 80, 2).astype(np.float32)).cuda()
    bits = torch.as_tensor(rng.randint(0, 2, (batch, 320, 2)).astype(np.int32)).cuda()
    out = model(x, bits)
    out["ce_mean"].backward()
    eng = RxEngine(model.dims(), batch, params=model.export_params(), train=True)
    eng.train_step(x, bits)
    torch.cuda.synchronize()
    assert torch.equal(out["input"], eng.x_norm) and torch.equal(out["fft_out"], eng.fft_out)
    assert torch.equal(out["output"], eng.prob)
    m1, m2 = read_metrics(out["metrics"]), eng.metrics()
    assert m1["conf"] == m2["conf"] and m1["ce_mean"] == m2["ce_mean"]
    assert abs(float(out["tx_power"]) - m2["tx_power"]) <= 1e-6 * m2["tx_power"]
    g = eng.get_grads()
    for n in g:
        ga = model.store.tensor(n).grad.cpu().numpy().reshape(g[n].shape)
        assert relerr(ga, g[n]) <= 2e-6, n
    # inference call: probabilities only, same values
    with torch.no_grad():
        assert torch.equal(model(x), eng.prob)
