"""OFDM frame grid + NumPy transmitter: the input generator / label layout either side of the
hot path (host-side mirror of dev/py/ofdm.py:198-380 -- class name and attribute names kept so
model code written against the reference's ``ofdmobj`` keeps working).

Pinned by tests/test_golden_substrate.py against vectors produced by the reference itself.
"""
from __future__ import annotations

import numpy as np

# LTE downlink numerology: FFT size -> (sample rate [Hz], resource blocks)   (dev/py/ofdm.py:173-194)
_LTE_DL = {64: (0.96e6, 4), 128: (1.92e6, 8), 256: (3.84e6, 15), 512: (7.68e6, 25),
           1024: (15.36e6, 50), 1536: (23.04e6, 75), 2048: (30.72e6, 100)}

_PILOT_VALUE = 3 + 3j
_QAM8_SCALE = 4.2426 / 3.1623          # |3+3i| / |3+1i|  (dev/py/ofdm.py:66-77)


def get_lte_dl_cfg(nfft: int):
    if nfft not in _LTE_DL:
        raise AssertionError("unsupported FFT size %r" % (nfft,))
    return _LTE_DL[nfft]


def const_map(nbits: int = 1) -> np.ndarray:
    """Constellation table indexed by the MSB-first symbol value (dev/py/ofdm.py:121-153).

    BPSK +-3*sqrt(2); QPSK / 8-QAM / 16-QAM on the {+-1,+-3} lattice with the reference's
    (non-Gray-standard) bit assignment: for 16-QAM the first two bits pick the imaginary level,
    the last two the real level; for 8-QAM the first bit picks the sign of the imaginary part.
    """
    if not 0 < nbits < 5:
        raise AssertionError("nbits must be 1..4")
    level = {(0, 0): -3.0, (1, 0): -1.0, (0, 1): 3.0, (1, 1): 1.0}      # (lsb-side pair) -> amplitude
    table = np.empty(2 ** nbits, dtype=np.complex64)
    for idx in range(2 ** nbits):
        b = [(idx >> (nbits - 1 - j)) & 1 for j in range(nbits)]         # b[0] is the MSB
        if nbits == 1:
            v = complex(4.24264 if b[0] else -4.24264, 0.0)
        elif nbits == 2:
            v = complex(3.0 if b[1] else -3.0, -3.0 if b[0] else 3.0)
        elif nbits == 3:
            v = complex(level[(b[1], b[2])], -1.0 if b[0] else 1.0) * _QAM8_SCALE
        else:
            v = complex(level[(b[2], b[3])], -level[(b[0], b[1])])
        table[idx] = v
    return table


class ofdm_tx:
    """LTE-like resource grid of one frame (``nsymbol`` OFDM symbols x ``nfft`` subcarriers).

    Attributes follow the reference object: K, CP, P, G, DC, Fs, nRB, nSymbol, pilotCarriers,
    dataCarriers, pilotSc, dataSc (flat ``symbol*K + carrier`` indices, sorted), frame_size,
    pilot_size.  Only the default ``pilot='lte'`` pattern of the sweep driver is provided
    (pilots on symbols 0 and 4, the second set shifted by three effective carriers).
    """

    def __init__(self, FLAGS):
        self.nSymbol = int(FLAGS.nsymbol)
        self.K = int(FLAGS.nfft)
        self.CP = int(np.around(self.K * (0.25 if FLAGS.longcp else 0.07)))
        self.Fs, self.nRB = get_lte_dl_cfg(self.K)
        self.DC = 2
        pilot = getattr(FLAGS, "pilot", "lte")
        if pilot != "lte":
            raise ValueError("Unsupported pilot type %s (only the driver default 'lte' is built)." % pilot)
        if self.nSymbol != 7:
            raise AssertionError("the LTE pilot pattern needs 7 symbols per frame")
        self.P = 2 * self.nRB
        self.G = self.K - self.DC - 12 * self.nRB
        self.pilotValue = _PILOT_VALUE
        self.guardValue = 0
        self.nbits = int(FLAGS.nbits)

        K, S = self.K, self.nSymbol
        self.allCarriers = np.arange(K)
        self.DCCarriers = np.array([K // 2 - 1, K // 2], dtype=np.int32)
        eff = np.arange(self.G // 2, K - self.G // 2)
        self.effecCarriers = eff[~np.isin(eff, self.DCCarriers)]
        n_eff = len(self.effecCarriers)
        step = int(np.ceil(float(n_eff) / self.P))
        self.pilot_loc = np.arange(0, n_eff, step)
        self.pilotCarriers = self.effecCarriers[self.pilot_loc]
        self.guardCarriers = np.setdiff1d(self.allCarriers, self.effecCarriers)
        self.dataCarriers = np.setdiff1d(self.effecCarriers, self.pilotCarriers)

        self.allSc = np.arange(K * S)
        self.effecSc = (np.arange(S)[:, None] * K + self.effecCarriers[None, :]).reshape(-1)
        first = self.effecCarriers[np.sort(self.pilot_loc % n_eff)]
        second = self.effecCarriers[np.sort((self.pilot_loc + 3) % n_eff)] + 4 * K
        self.pilotSc = np.sort(np.concatenate([first, second]))
        self.guardSc = np.setdiff1d(self.allSc, self.effecSc)
        self.dataSc = np.setdiff1d(self.effecSc, self.pilotSc)
        self.frame_size = len(self.dataSc)
        self.pilot_size = len(self.pilotSc)

    # -------------------------------------------------------------------------------------
    def symbols_from_bits(self, bits: np.ndarray) -> np.ndarray:
        """bits [..., nbits] (MSB first) -> constellation points (complex64)."""
        nbits = bits.shape[-1]
        weights = (1 << np.arange(nbits - 1, -1, -1)).astype(np.int64)
        return const_map(nbits).take(bits.astype(np.int64) @ weights)

    def ofdm_tx_frame_np(self, inputs: np.ndarray):
        """bits [n_frame, frame_size, nbits] -> (complex64 [n,S,K+CP], float32 [n,S,K+CP,2],
        pilot reference float32 [n,S,P,2])   (dev/py/ofdm.py:328-380)."""
        n_frame, frame_size, nbits = (int(v) for v in inputs.shape)
        if frame_size != self.frame_size:
            raise AssertionError("expected %d data cells per frame, got %d" % (self.frame_size, frame_size))
        if nbits >= 5:
            raise AssertionError("at most 4 bits per symbol")
        K, S = self.K, self.nSymbol
        grid = np.zeros((n_frame, S * K), dtype=np.complex64)
        grid[:, self.dataSc] = self.symbols_from_bits(inputs)
        grid[:, self.pilotSc] = self.pilotValue
        time = np.fft.ifft(grid.reshape(n_frame * S, K))               # single precision on complex64
        with_cp = np.concatenate([time[:, K - self.CP:], time], axis=1).reshape(n_frame, S, K + self.CP)
        real = np.stack([with_cp.real, with_cp.imag], axis=-1)
        pilot = 3.0 * np.ones((n_frame, S, self.P, 2), dtype=np.float32)
        return with_cp, real, pilot
