#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ from the reference itself.

Runs ONLY in the build container (needs /root/reference).  The reference's NumPy
substrate (dev/py/ofdm.py, radio.py, util.py) imports ``tensorflow`` at module top but
its NumPy functions never touch it, so a MagicMock stands in for the module
(SURVEY.md §8c / Appendix C).  Nothing of the reference's source travels: the outputs
are plain data (inputs + expected outputs) stored as .npz / .json.

    python tests/golden/make_golden.py

Fixtures written
    grid_tables.json        ofdm_tx grid config (ofdm.py:198-273) for nfft in {64,1024} x longcp
    const_maps.npz          const_map(1..4) (ofdm.py:121-153)
    tx_frames.npz           seeded bit_source (util.py:25-34) -> ofdm_tx_frame_np (ofdm.py:328-380)
    channels.npz            seeded rayleigh_chan_lte.run (radio.py:277-510), static + Doppler + mix
    awgn.npz                seeded AWGN_channel_np (radio.py:513-526)
    v1_index_manifest.json  variable names/shapes/dtypes parsed from test_v1/model/*.index
    lte_tap_interp.json     (written to dl_ofdm_amd/data/) the 3gpp/AM_*.csv tap-interpolation matrices
"""
import json
import os
import sys
import types
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
REF_PY = os.path.join(REF, "dev", "py")


def import_reference():
    tf = mock.MagicMock()
    tf.__version__ = "1.15.5"
    sys.modules["tensorflow"] = tf
    sys.path.insert(0, REF_PY)
    os.chdir(REF_PY)                      # radio.py loads ./3gpp/*.csv relative to cwd
    import ofdm, radio, util              # noqa: E401
    return ofdm, radio, util


def flags(**kw):
    base = dict(nsymbol=7, nfft=64, longcp=True, pilot="lte", npilot=8, nguard=8, nbits=2,
                channel="EPA")
    base.update(kw)
    return types.SimpleNamespace(**base)


# ---- TF checkpoint .index (LevelDB-style SSTable) reader --------------------------------
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _block_entries(buf, off, size):
    blk = buf[off:off + size]
    n_restarts = int.from_bytes(blk[-4:], "little")
    end = len(blk) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(blk, pos)
        non_shared, pos = _varint(blk, pos)
        vlen, pos = _varint(blk, pos)
        key = key[:shared] + blk[pos:pos + non_shared]
        pos += non_shared
        yield key, blk[pos:pos + vlen]
        pos += vlen


def _proto_fields(buf):
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        fno, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError(wt)
        yield fno, wt, v


def read_index(path):
    buf = open(path, "rb").read()
    footer = buf[-48:]
    _, p = _varint(footer, 0)
    _, p = _varint(footer, p)
    ioff, p = _varint(footer, p)
    isz, p = _varint(footer, p)
    out = {}
    for _, handle in _block_entries(buf, ioff, isz):
        doff, q = _varint(handle, 0)
        dsz, q = _varint(handle, q)
        for key, val in _block_entries(buf, doff, dsz):
            if not key:
                continue                                   # header entry
            ent = dict(dtype=None, shape=[], offset=0, size=0)
            for fno, wt, v in _proto_fields(val):
                if fno == 1:
                    ent["dtype"] = int(v)
                elif fno == 2:
                    for f2, _, dim in _proto_fields(v):
                        if f2 == 2:
                            sz = 0
                            for f3, _, d in _proto_fields(dim):
                                if f3 == 1:
                                    sz = int(d)
                            ent["shape"].append(sz)
                elif fno == 4:
                    ent["offset"] = int(v)
                elif fno == 5:
                    ent["size"] = int(v)
            out[key.decode()] = ent
    return out


def main():
    ofdm, radio, util = import_reference()
    os.makedirs(HERE, exist_ok=True)

    # 1. grid tables ----------------------------------------------------------------
    tables = {}
    for nfft in (64, 1024):
        for longcp in (True, False):
            o = ofdm.ofdm_tx(flags(nfft=nfft, longcp=longcp))
            tables["nfft%d_longcp%d" % (nfft, int(longcp))] = dict(
                K=int(o.K), CP=int(o.CP), P=int(o.P), G=int(o.G), DC=int(o.DC),
                Fs=float(o.Fs), nRB=int(o.nRB), nSymbol=int(o.nSymbol),
                frame_size=int(o.frame_size), pilot_size=int(o.pilot_size),
                pilotCarriers=o.pilotCarriers.tolist(), dataCarriers=o.dataCarriers.tolist(),
                pilotSc=o.pilotSc.tolist(), dataSc=o.dataSc.tolist(),
                guardSc_len=int(len(o.guardSc)))
    with open(os.path.join(HERE, "grid_tables.json"), "w") as f:
        json.dump(tables, f)

    # 2. constellation maps -----------------------------------------------------------
    np.savez(os.path.join(HERE, "const_maps.npz"),
             **{"nbits%d" % b: ofdm.const_map(b) for b in (1, 2, 3, 4)})

    # 3. transmitter ------------------------------------------------------------------
    tx = {}
    for nbits in (1, 2, 3, 4):
        o = ofdm.ofdm_tx(flags(nbits=nbits))
        np.random.seed(100 + nbits)
        bits = util.bit_source(nbits, o.frame_size, 3)
        cpx, real, pilot = o.ofdm_tx_frame_np(bits)
        tx["n64_b%d_bits" % nbits] = bits.astype(np.int8)
        tx["n64_b%d_cpx" % nbits] = cpx
        tx["n64_b%d_real" % nbits] = real
    o = ofdm.ofdm_tx(flags(nbits=2, longcp=False))
    np.random.seed(7)
    bits = util.bit_source(2, o.frame_size, 2)
    cpx, real, _ = o.ofdm_tx_frame_np(bits)
    tx["n64s_b2_bits"], tx["n64s_b2_cpx"], tx["n64s_b2_real"] = bits.astype(np.int8), cpx, real
    o = ofdm.ofdm_tx(flags(nbits=2, nfft=1024, longcp=False))      # BASELINE config 4 (CP=72)
    np.random.seed(8)
    bits = util.bit_source(2, o.frame_size, 1)
    cpx, real, _ = o.ofdm_tx_frame_np(bits)
    tx["n1024_b2_bits"], tx["n1024_b2_cpx"], tx["n1024_b2_real"] = bits.astype(np.int8), cpx, real
    np.savez_compressed(os.path.join(HERE, "tx_frames.npz"), **tx)

    # 4. channels ---------------------------------------------------------------------
    ch = {}
    o = ofdm.ofdm_tx(flags(nbits=2))
    np.random.seed(11)
    bits = util.bit_source(2, o.frame_size, 6)
    cpx, _, _ = o.ofdm_tx_frame_np(bits)
    ch["tx_cpx"] = cpx
    for i, name in enumerate(["AWGN", "Flat", "EPA", "EVA", "ETU", "Custom"]):
        fad = radio.rayleigh_chan_lte(flags(channel=name), o.Fs)
        np.random.seed(200 + i)
        y, H = fad.run(cpx)
        ch["static_%s_y" % name], ch["static_%s_H" % name] = y, H
    for i, name in enumerate(["EPA", "EVA", "ETU"]):
        fad = radio.rayleigh_chan_lte(flags(channel=name), o.Fs, mobile=True)
        np.random.seed(300 + i)
        y, H = fad.run(cpx[:3])
        ch["doppler_%s_y" % name], ch["doppler_%s_H" % name] = y, H
    fad = radio.rayleigh_chan_lte(flags(channel="mixRayleigh"), o.Fs, mobile=True, mix=True)
    np.random.seed(400)
    y, H = fad.run(cpx)
    ch["mix_y"], ch["mix_H"] = y, H
    np.savez_compressed(os.path.join(HERE, "channels.npz"), **ch)

    # 5. AWGN -------------------------------------------------------------------------
    np.random.seed(500)
    xin = ch["static_EPA_y"]
    snr = np.array([[-10.0], [0.0], [5.0], [10.0], [20.0], [30.0]])
    yout, npow = radio.AWGN_channel_np(xin, snr)
    np.savez_compressed(os.path.join(HERE, "awgn.npz"), x=xin, snr=snr, y=yout,
                        noise_power=np.float64(npow))

    # 6. v1 checkpoint index manifest --------------------------------------------------
    man = {}
    mdir = os.path.join(REF, "test_v1", "model")
    for fn in sorted(os.listdir(mdir)):
        if fn.endswith(".index"):
            man[fn[:-6]] = read_index(os.path.join(mdir, fn))
    with open(os.path.join(HERE, "v1_index_manifest.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)

    # 7. LTE tap-interpolation matrices (data the channel model needs) ----------------
    am = {}
    for name in ("EPA", "EVA", "ETU", "Custom"):
        a = np.genfromtxt(os.path.join(REF_PY, "3gpp", "AM_%s.csv" % name), delimiter=",")
        am[name.lower()] = a.tolist()
    ddir = os.path.join(REPO, "dl_ofdm_amd", "data")
    os.makedirs(ddir, exist_ok=True)
    with open(os.path.join(ddir, "lte_tap_interp.json"), "w") as f:
        json.dump(dict(source="MATLAB rayleighchan alphaMatrix for LTE EPA/EVA/ETU/Custom at the "
                              "nfft=64 sample rate (reference dev/py/3gpp/AM_*.csv)",
                       matrices=am), f)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
