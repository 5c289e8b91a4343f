"""dl_ofdm_amd -- MI355X-native DCCN OFDM receiver hot path.

The compute lives in ``dl_ofdm_amd/lib/libdccn.so`` (hand-written gfx950 HIP kernels behind the
C ABI of ``include/dccn.h``); this package is the host-side mirror of the reference's layer /
model / harness interface (dev/py/complex.py, model.py, ofdmreceiver_np.py).
"""
__version__ = "0.1.0"
