"""Chain groups (include/dccn.h dccn_eq_train_step_grouped, dl_ofdm_amd/equalizer_group.py): several equaliser training chains --
the reference driver's per-modulation jobs, dev/py/run_local_ofdm.py:61-118, loop dev/py/ofdmreceiver_np_mp.py:394-466 -- carried
by ONE launch sequence.  The bar is bitwise: a chain trained inside a group ends with exactly the parameters, Adam slots, step
counter and per-epoch history it gets from receiver_mp.train on its own."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _flags(nbits, tmp, **kw):
    from dl_ofdm_amd import receiver_mp as H
    base = dict(nbits=nbits, nfilter=64, channel="mixRayleigh", device_data=True, seed=10 + nbits, max_epoch_num=3, early_stop=200,
                msg_length=7 * 73 * 5, eval_frames=128, token="grp%d" % nbits, save_dir=os.path.join(str(tmp), "ck%d/" % nbits))
    base.update(kw)
    return H.Flags(**base)


def _rx(F, seed):
    from dl_ofdm_amd import ofdm, receiver as R
    from dl_ofdm_amd.engine import glorot_init
    return glorot_init(R.rx_dims(F, ofdm.ofdm_tx(F)), seed)


def _state(tr):
    return [t.clone() for t in (tr.params, tr.adam_m, tr.adam_v, tr.adam_state, tr.grads)]


@pytest.mark.parametrize("mods", [(1, 2, 3, 4), (2, 2), (4, 3, 4, 1, 2, 2, 3, 1)])
def test_grouped_chains_equal_their_solo_runs_bit_for_bit(tmp_path, mods):
    """G chains of mixed modulations through dccn_gen_static_frames_grouped / dccn_eq_train_step_grouped /
    dccn_eq_monitor_accumulate_grouped (three C calls per group step) against receiver_mp.train per chain: three epochs of five
    73-frame steps + the per-epoch evaluation; identical arenas, Adam state and history."""
    from dl_ofdm_amd import receiver_mp as H
    from dl_ofdm_amd.equalizer_group import train_group
    fl = [_flags(nb, tmp_path / "g", seed=10 + 7 * i + nb, token="g%d_%d" % (i, nb)) for i, nb in enumerate(mods)]
    rx = [_rx(F, 3 + i) for i, F in enumerate(fl)]
    grouped = train_group(fl, rx, verbose=False)
    torch.cuda.synchronize()
    for i, (F, r) in enumerate(zip(fl, rx)):
        Fs = _flags(F.nbits, tmp_path / "s", seed=F.seed, token=F.token)
        solo = H.train(Fs, verbose=False, run_test=False, rx_params=r)
        torch.cuda.synchronize()
        a, b = _state(grouped[i]["trainer"]), _state(solo["trainer"])
        for name, x, y in zip(("params", "adam_m", "adam_v", "adam_state", "grads"), a, b):
            assert torch.equal(x, y), (i, F.nbits, name, float((x - y).abs().max()))
        assert grouped[i]["history"] == solo["history"], (i, F.nbits)
        assert len(solo["history"]) == 3 and float(a[3][0]) == 15.0
        za, zb = np.load(grouped[i]["best_path"] + ".npz"), np.load(solo["best_path"] + ".npz")
        for k in za.files:
            if k != "__flags__":
                assert np.array_equal(za[k], zb[k]), k


def test_chains_leave_the_group_at_their_own_early_stop(tmp_path):
    """chains stop when THEIR epoch budget / early stopping says so; the others go on in a smaller group -- still bit for bit"""
    from dl_ofdm_amd import receiver_mp as H
    from dl_ofdm_amd.equalizer_group import train_group
    epochs = {1: 2, 2: 4, 4: 3}
    fl = [_flags(nb, tmp_path / "g", max_epoch_num=epochs[nb]) for nb in (1, 2, 4)]
    rx = [_rx(F, 5) for F in fl]
    grouped = train_group(fl, rx)
    for F, r, g in zip(fl, rx, grouped):
        solo = H.train(_flags(F.nbits, tmp_path / "s", max_epoch_num=F.max_epoch_num), verbose=False, run_test=False, rx_params=r)
        assert len(g["history"]) == epochs[F.nbits] and g["history"] == solo["history"]
        assert torch.equal(g["trainer"].params, solo["trainer"].params)
        assert torch.equal(g["trainer"].adam_state, solo["trainer"].adam_state)


def test_grouped_step_refuses_buffers_that_break_the_layout_contract(tmp_path):
    """every pointer of chain g must lie at one offset from chain 0's: a chain whose label buffer sits elsewhere is refused with
    DCCN_ERR_INVALID_ARG before anything is launched; shapes that differ in more than the modulation likewise"""
    from dl_ofdm_amd import _lib
    from dl_ofdm_amd.equalizer_group import EqualizerChainGroup, _ptr_array
    fl = [_flags(nb, tmp_path) for nb in (2, 4)]
    grp = EqualizerChainGroup(fl, [_rx(F, 1) for F in fl])
    lib = grp.lib
    assert int(lib.dccn_chain_group_max()) == 8 and int(lib.dccn_eq_group_supported(C.byref(grp.chains[0].pl.shape))) == 1
    act = grp.chains
    for c in act:
        c.begin_epoch()
    grp.step(act, 0)
    torch.cuda.synchronize()
    before = [c.tr.params.clone() for c in act]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    good = [c.loop.pls[1].pipe_buffers[1] for c in act]
    vals = {f: getattr(good[1], f) for f, _ in _lib.EqBuffers._fields_}
    stray = torch.zeros_like(act[1].loop.pls[1].bits)
    vals["bits"] = stray.data_ptr()
    bad = _lib.EqBuffers(*[vals[f] for f, _ in _lib.EqBuffers._fields_])
    shapes = _ptr_array([c.loop.pls[1].shape for c in act])
    rc = lib.dccn_eq_train_step_grouped(2, shapes, _ptr_array([good[0], bad]), grp.hp, st)
    assert rc == -1
    other = _lib.EqShape.from_buffer_copy(act[1].loop.pls[1].shape)
    other.batch = 64
    rc2 = lib.dccn_eq_train_step_grouped(2, _ptr_array([act[0].loop.pls[1].shape, other]), _ptr_array(good), grp.hp, st)
    assert rc2 == -1
    big = _lib.EqShape.from_buffer_copy(act[0].loop.pls[1].shape)
    big.batch = 1170
    assert int(lib.dccn_eq_group_supported(C.byref(big))) == 0
    torch.cuda.synchronize()
    for c, p in zip(act, before):
        assert torch.equal(c.tr.params, p)
