"""Equaliser transfer-learning harness: trained basic receiver -> train ``Equalizer/*`` over a fading
channel -> cross-channel SNR sweep -> CSV.  Host-side mirror of dev/py/ofdmreceiver_np_mp.py:

  * flags (:32-59; same names and defaults, except ``opt`` -- see below)
  * graph surgery (:264-306) = :class:`dl_ofdm_amd.equalizer.EqualizerTrainer` (frozen receiver +
    trainable equaliser on the flat Adam arena)
  * epoch schedule (:381-466): per epoch ``msg_length // nsymbol`` frames, per-frame training SNR drawn
    from linspace(0, 27, 10) with the reference's probabilities, fading + AWGN on the host substrate,
    ``batch_size // nsymbol`` frames per step, best-train-loss checkpoint, early stop
  * final test (:62-104): channels ETU/EVA/EPA/Flat/Custom x SNR -10..30 step 5 x 30 000 frames ->
    ``Test_DCCN_<token>_Equalizer<opt>_<channel>_test_chan_<chan>[_mobile].csv``; the (channel, SNR)
    points are sharded over ranks exactly like the basic sweep (sweep.py).

Only ``equalizer_ofdm`` is built (opt 0, 9, 10 -- ofdmreceiver_np_mp.py:285,301-304); the ablation
variants (opt 1-7) are outside the scope table in DESIGN.md and raise NotImplementedError.  The
reference's default ``opt=3`` is one of those, so the default here is 0.

    python -m dl_ofdm_amd.receiver_mp --token=OFDM_QPSK --nbits=2 --channel=EPA --nfilter=64 --max_epoch_num=20
"""
from __future__ import annotations

import argparse
import copy
import os
import time
from dataclasses import asdict, dataclass
from typing import Dict, Optional

import numpy as np

from . import ofdm, radio, sweep, util
from .engine import PARAM_NAMES
from .receiver import _bool

TRAIN_SNR_GRID = np.linspace(0, 27, 10, dtype=np.float32)                      # :387 aa_milne_arr
TRAIN_SNR_PROB = [0.01, 0.01, 0.02, 0.02, 0.02, 0.02, 0.1, 0.5, 0.2, 0.1]      # :407
TEST_CHANNELS = ("ETU", "EVA", "EPA", "Flat", "Custom")                        # :74


@dataclass
class Flags:
    """tf.app.flags of ofdmreceiver_np_mp.py:32-59."""
    save_dir: str = "./output/"
    nbits: int = 1
    msg_length: int = 100800
    batch_size: int = 512
    max_epoch_num: int = 5000
    seed: int = 1
    nfft: int = 64
    nsymbol: int = 7
    npilot: int = 8
    nguard: int = 8
    nfilter: int = 80
    SNR: float = 30.0
    SNR2: float = 30.0
    early_stop: int = 400
    ofdm: bool = True
    pilot: str = "lte"
    channel: str = "EPA"
    cp: bool = True
    longcp: bool = True
    load_model: bool = True
    split: float = 1.0
    token: str = "OFDM"
    opt: int = 0
    mobile: bool = False
    init_learning: float = 0.001
    test: bool = False
    # additions of this implementation
    test_frames: int = 30000        # frames per sweep point (:72)
    eval_frames: int = 1024         # per-epoch evaluation batch (:430)
    snr_lo: int = -10
    snr_hi: int = 30
    snr_step: int = 5               # :81
    device_data: bool = False       # generate bits/frames/channel/noise on the GPU (datagen.py)
    align_window: bool = False      # extension: delay generated frames by the channel's centre-tap advance (datagen.py)
    pipeline_norm: Optional[bool] = None  # device_data: step i normalises batch i+1 on its optimizer launch (None: when the library can)
    overlap_generator: bool = False  # device_data: generate batch i+1 on a second stream while step i runs (same results, not faster)
    step_graph: bool = False         # device_data: replay the equaliser step as a hipGraph instead of 21 eager launches (same results)
    virtual_next: bool = True        # ... and the pipelined loop never writes x for any batch but an epoch's first (x_next_virtual)
    monitor_rides: bool = True       # device_data, pipelined loop: the per-step monitors are a job of the step's optimizer launch
    generator_rides: bool = True     # ... and that launch's workgroups ride on the step that normalises the batch (no launch of its own)
    fused_generator: bool = True     # device_data, static channels: one generator launch per batch (datagen.FusedStaticGen; same
                                     # draws, transmitted frames equal to rounding) instead of the launch-per-stage chain
    tf_checkpoint: bool = False     # also write the tf.train.Saver bundle (.index/.data-00000-of-00001)


def parse_flags(argv=None) -> Flags:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for k, v in asdict(Flags()).items():
        ap.add_argument("--" + k, type=_bool if isinstance(v, bool) else type(v), default=v)
    return Flags(**vars(ap.parse_args(argv)))


class RayleighChanParallel:
    """ofdmreceiver_np_mp.py:191-223 fans frames out over a multiprocessing pool of per-frame simulators;
    :class:`dl_ofdm_amd.radio.rayleigh_chan_lte` is already batched over frames, so this is the same
    ``run(iq_tx_cmpx) -> (rx [n,S,n_sc,2], H [n,S,nfft])`` interface over one vectorised call."""

    def __init__(self, flags, sample_rate=0.96e6, mobile=False, mix=False):
        self.obj = radio.rayleigh_chan_lte(flags, sample_rate, mobile, mix)

    def run(self, iq_tx_cmpx):
        return self.obj.run(iq_tx_cmpx)


def save_model_name(FLAGS) -> str:
    """:332-335"""
    if FLAGS.opt == 0:
        return FLAGS.token + "_Equalizer_" + FLAGS.channel
    return FLAGS.token + "_Equalizer%d_" % FLAGS.opt + FLAGS.channel


def load_rx_params(FLAGS) -> Dict[str, np.ndarray]:
    """the trained basic receiver written by dl_ofdm_amd.receiver (``<save_dir>/<token>.npz``)."""
    from .receiver import read_checkpoint_file
    stem = os.path.join(FLAGS.save_dir, FLAGS.token)
    if not os.path.exists(stem + ".npz") and not os.path.exists(stem + ".index"):
        raise FileNotFoundError("basic-receiver checkpoint %s(.npz|.index) not found: train it first with "
                                "`python -m dl_ofdm_amd.receiver --token=%s ...`" % (stem, FLAGS.token))
    # the reference restores the basic receiver's graph and takes its tensors by name (:264-285); same contract here:
    # the checkpoint goes through the graph-name API, the equaliser is then grafted in front of `input:0`
    from .session import Session, load_model_np
    sess = Session(device="cuda", seed=FLAGS.seed)
    load_model_np(stem, sess)
    for name in ("bits_in:0", "tx_ofdm:0", "SNR:0", "tx_signal:0", "tx_power:0", "iq_tx:0", "iq_rx:0", "input:0",
                 "ce_mean:0", "noise_power:0", "log_ber:0", "linear_ber:0", "conf_matrix:0"):
        sess.get_tensor_by_name(name)
    if sess.dims.nbits != FLAGS.nbits or sess.dims.F != FLAGS.nfilter or sess.dims.S != FLAGS.nsymbol:
        raise ValueError("basic receiver %s was trained with nbits=%d nfilter=%d nsymbol=%d, flags say %d/%d/%d"
                         % (stem, sess.dims.nbits, sess.dims.F, sess.dims.S, FLAGS.nbits, FLAGS.nfilter, FLAGS.nsymbol))
    return dict(sess.params)


def make_batch(FLAGS, ofdmobj, fading, n_frames: int, snr_db):
    """bits -> OFDM frames -> fading -> AWGN (:405-413); also returns the true channel response."""
    ys = util.bit_source(FLAGS.nbits, ofdmobj.frame_size, n_frames)
    iq_cpx, _, _ = ofdmobj.ofdm_tx_frame_np(ys)
    xs, chan = fading.run(iq_cpx)
    snr = snr_db * np.ones((n_frames, 1)) if np.isscalar(snr_db) else snr_db
    xs, noise_pwr = radio.AWGN_channel_np(xs, snr)
    return xs.astype(np.float32), ys.astype(np.int32), chan, noise_pwr


def test_model_cross(FLAGS, trainer, ofdmobj, rank: int = 0, world: int = 1, out_dir: str = ".",
                     verbose: bool = True, channels=TEST_CHANNELS):
    """:62-104.  Returns {test_chan: (snrs, ber, loss, csvfile)}; rank 0 writes the CSV files."""
    snrs = list(range(FLAGS.snr_lo, FLAGS.snr_hi + 1, FLAGS.snr_step))
    pts = sweep.make_points([FLAGS.nbits], list(channels), snrs, base_seed=FLAGS.seed)
    fadings = {}

    on_device = bool(getattr(FLAGS, "device_data", False)) and trainer.fused_ok

    def evaluate(p):
        if p.channel not in fadings:
            fl = copy.deepcopy(FLAGS)
            fl.channel = p.channel
            if on_device:
                from .datagen import DeviceDataGen
                fadings[p.channel] = DeviceDataGen(fl, ofdmobj, device=trainer.device, seed=p.seed,
                                                   mobile=FLAGS.mobile)
                fadings[p.channel].want_noise_power = False
            else:
                fadings[p.channel] = RayleighChanParallel(fl, ofdmobj.Fs, mobile=FLAGS.mobile)
        if on_device:
            gen, pl = fadings[p.channel], trainer.resident(FLAGS.test_frames)
            gen.seed, gen.offset = p.seed, 0
            gen.make_batch(FLAGS.test_frames, p.snr_db, out_x=pl.x, out_bits=pl.bits)
            pl.run(False)
            m = trainer._metrics(pl.metrics_buf, pl.tx_power)
        else:
            np.random.seed(p.seed)
            xs, ys, _, _ = make_batch(FLAGS, ofdmobj, fadings[p.channel], FLAGS.test_frames, p.snr_db)
            m = trainer.eval_step(xs, ys)
        if verbose:
            print("Test in %s: SNR: %.2f, BER: %.8f, Loss: %f" % (p.channel, p.snr_db, m["berlin"], m["ce_mean"]))
        c = m["conf"]
        return [c[0][0], c[0][1], c[1][0], c[1][1], m["ce_sum"], m["count"]]

    table = sweep.run_sweep(pts, evaluate, rank, world, device=trainer.device)
    ber, loss = sweep.ber_loss(table)
    out = {}
    for ci, ch in enumerate(channels):
        sl = slice(ci * len(snrs), (ci + 1) * len(snrs))
        name = "Test_DCCN_%s_test_chan_%s%s.csv" % (FLAGS.token + "_Equalizer%d_" % FLAGS.opt + FLAGS.channel, ch,
                                                   "_mobile" if FLAGS.mobile else "")
        path = os.path.join(out_dir, name)
        if rank == 0:
            sweep.write_csv(path, snrs, ber[sl], loss[sl])
        out[ch] = (snrs, ber[sl], loss[sl], path)
    return out


def save_checkpoint(path: str, trainer, FLAGS):
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    out = trainer.state_dict_tf()
    if getattr(FLAGS, "tf_checkpoint", False):
        from . import tf_bundle
        tf_bundle.write_checkpoint(path[:-4] if path.endswith(".npz") else path, tf_bundle.eq_to_tf(out))
    out["__flags__"] = np.array(repr(asdict(FLAGS)))
    np.savez(path if path.endswith(".npz") else path + ".npz", **out)
    return path


def load_checkpoint(path: str, trainer, with_optimizer: bool = True):
    stem = path[:-4] if path.endswith(".npz") else path
    if not os.path.exists(stem + ".npz") and os.path.exists(stem + ".index"):
        from . import tf_bundle
        z = tf_bundle.eq_from_tf(tf_bundle.read_checkpoint(stem))
    else:
        z = np.load(stem + ".npz", allow_pickle=False)
    trainer.load_state_dict_tf(z, with_optimizer)


def train(FLAGS, device="cuda", verbose: bool = True, run_test: bool = True, rx_params=None):
    from .equalizer import EqualizerTrainer
    ofdmobj = ofdm.ofdm_tx(FLAGS)
    frame_cnt = FLAGS.msg_length // FLAGS.nsymbol
    np.random.seed(FLAGS.seed)
    rx_params = rx_params if rx_params is not None else load_rx_params(FLAGS)
    trainer = EqualizerTrainer(FLAGS, ofdmobj, rx_params, device=device, seed=FLAGS.seed).pin_tuning()
    batch_size = FLAGS.batch_size // FLAGS.nsymbol                     # :341
    fading0 = RayleighChanParallel(FLAGS, ofdmobj.Fs, mobile=False)    # :389
    fading1 = RayleighChanParallel(FLAGS, ofdmobj.Fs, mobile=True, mix=True) if FLAGS.mobile else None
    phase2 = True                                                      # :393
    loss_min, epoch_min, best_path, history = 100.0, 0, "", []
    on_device = bool(FLAGS.device_data) and trainer.fused_ok
    if on_device:
        return _train_on_device(FLAGS, ofdmobj, trainer, batch_size, frame_cnt, verbose, run_test)
    for epoch in range(FLAGS.max_epoch_num):
        np.random.seed(FLAGS.seed + 1000003 * (epoch + 1))              # reference: int(time.time()) + epoch
        train_snr = np.random.choice(TRAIN_SNR_GRID, [frame_cnt, 1], p=TRAIN_SNR_PROB)     # :407
        fading = fading1 if (phase2 and FLAGS.mobile) else fading0
        ys = util.bit_source(FLAGS.nbits, ofdmobj.frame_size, frame_cnt)
        iq_cpx, _, _ = ofdmobj.ofdm_tx_frame_np(ys)
        xs, chan = fading.run(iq_cpx)
        xs, noise_pwr = radio.AWGN_channel_np(xs, train_snr)
        xs, ys = xs.astype(np.float32), ys.astype(np.int32)
        losses, pwrs, bers, rmss = [], [], [], []
        for i in range(frame_cnt // batch_size):
            sl = slice(i * batch_size, (i + 1) * batch_size)
            m = trainer.train_step(xs[sl], ys[sl], chan[sl])
            losses.append(m["ce_mean"]); pwrs.append(m["tx_power"]); bers.append(m["berlin"]); rmss.append(m["chan_rms"])
        train_loss_epoch = float(np.mean(losses))
        test_snr = np.random.choice(TRAIN_SNR_GRID, [FLAGS.eval_frames, 1], p=TRAIN_SNR_PROB)   # :438
        txs, tys, _, _ = make_batch(FLAGS, ofdmobj, fading, FLAGS.eval_frames, test_snr)
        em = trainer.eval_step(txs, tys)
        history.append(dict(epoch=epoch, train_loss=train_loss_epoch, train_ber=float(np.mean(bers)),
                            chan_rms=float(np.mean(rmss)), test_loss=em["ce_mean"], test_ber=em["berlin"]))
        if verbose:
            print("Epoch: %d  Train Loss: %f  Tx Power: %f  Noise Power: %f  SNR MSE: %f | Test Loss: %f  Test BER: %.8f"
                  % (epoch, train_loss_epoch, float(np.mean(pwrs)), float(np.mean(noise_pwr)), float(np.mean(rmss)),
                     em["ce_mean"], em["berlin"]))
        if train_loss_epoch < loss_min:                                  # :456-459
            epoch_min, loss_min = epoch, train_loss_epoch
            best_path = save_checkpoint(os.path.join(FLAGS.save_dir, save_model_name(FLAGS)), trainer, FLAGS)
        if epoch - FLAGS.early_stop > epoch_min:                         # :460-466
            break
    if verbose:
        print("Training Done!, Best model saved to\n%s" % best_path)
    result = dict(history=history, best_path=best_path, trainer=trainer)
    if run_test and best_path:
        load_checkpoint(best_path, trainer, with_optimizer=False)
        result["sweep"] = test_model_cross(FLAGS, trainer, ofdmobj, verbose=verbose)
    return result


class DeviceEpochLoop:
    """One training step of the on-device epoch loop as four library calls and nothing else on the host: transmit, channel
    (+ AWGN at this step's row of the epoch's SNR table, already on the device), the fused equaliser step, and ONE monitor
    launch that computes `chan_rms` (ofdmreceiver_np_mp.py:245, 325-333) and adds the step's scalars onto the epoch
    accumulators.  Round 3 did the last part with ~25 framework launches and a host-to-device copy per step, which made the
    loop host-bound at twice the fused step's own time (tools/eqloop.py).

    ``overlap`` (built, bitwise-tested, measured, OFF by default): two resident buffer sets (input, labels, channel truth,
    noise power, step workspace) and the generator on its own HIP stream -- batch i+1 is produced into the other set while
    step i runs; two events per set order producer and consumer.  Same batches, same step: the trained model is bit-identical
    to the one-stream loop (tests/test_gpu_equalizer.py) -- but not faster: 0.2222 vs 0.2172 ms per step (tools/eqloop.py);
    the two cross-stream edges per step cost what hiding the generator's 26 us gains.

    ``pipeline`` (default: on when the library honours ``dccn_eq_buffers.x_next`` for this shape): two twin plans (own input
    and labels, shared workspace and outputs) hold consecutive batches; batch i+1 is generated BEFORE step i is issued, and
    step i normalises it on leading workgroups of its optimizer launch, so every step but the first of an epoch starts at
    the layer norm -- one launch (6 us) off a 21-launch chain.  Same kernels on the same data: bit-identical training."""

    def __init__(self, FLAGS, ofdmobj, trainer, gen, pl, steps: int, overlap: bool = False, pipeline: Optional[bool] = None,
                 arena=None):
        """arena: the chain's :class:`~dl_ofdm_amd.arena.ChainArena` (chain groups, equalizer_group.py): every buffer a launch of
        the loop touches is placed there"""
        import ctypes as C
        import torch
        from . import arena as A
        from .equalizer import _FusedPlan
        self.C, self.torch = C, torch
        self.F, self.o, self.tr, self.gen, self.steps = FLAGS, ofdmobj, trainer, gen, int(steps)
        self.arena = arena
        dev, B = trainer.device, pl.batch
        self.overlap = bool(overlap)
        if pipeline is None:
            pipeline = not self.overlap and bool(trainer.lib.dccn_eq_norm_rides(C.byref(pl.shape)))
        self.pipeline = bool(pipeline) and not self.overlap
        if self.pipeline:
            self.pls = [pl, _FusedPlan(trainer, B, twin_of=pl, arena=arena)]
            self.pls[0].pipe_with(self.pls[1], 0)
            self.pls[1].pipe_with(self.pls[0], 1)
        else:
            self.pls = [pl, _FusedPlan(trainer, B)] if self.overlap else [pl]
        self.pl = pl
        self.per_symbol = 1 if (gen.doppler or gen.mixed) else 0
        hshape = (B, FLAGS.nsymbol, ofdmobj.K, 2) if self.per_symbol else (B, ofdmobj.K, 2)
        n = len(self.pls)
        self.H = [A.empty(arena, *hshape, dtype=torch.float32, device=dev) for _ in range(n)]
        self.npow = [A.zeros(arena, 1, dtype=torch.float32, device=dev) for _ in range(n)]
        self.acc = A.zeros(arena, 5, dtype=torch.float32, device=dev)
        self.snr = A.zeros(arena, self.steps, B, dtype=torch.float32, device=dev)
        self.snr_rows = [self.snr[i] for i in range(self.steps)]
        # static channels (one profile, or mixRayleigh's frame-interleaved profiles without Doppler frames): the whole
        # generator chain of a batch is ONE launch + the launch that forms x (datagen.FusedStaticGen) instead of 5 to 12
        self.fg = None
        if getattr(FLAGS, "fused_generator", True):
            from .datagen import FusedStaticGen
            if FusedStaticGen.supported(gen):
                self.fg = FusedStaticGen(gen, B, 0.0, want_noise_power=gen.want_noise_power, arena=arena)
                if self.fg.npow is not None:
                    self.npow = [self.fg.npow[k] for k in range(n)]
        # ... and in the pipelined loop the second of those launches goes too: the optimizer launch of step i reads batch i + 1
        # as (y, noise, power partials) -- dccn_eq_buffers.x_next_virtual; only an epoch's first batch is materialised
        self.virt = None
        if self.fg is not None and self.pipeline and getattr(FLAGS, "virtual_next", True):
            from . import _lib
            self.virt = []
            for q in range(2):                                       # plan q normalises plan q ^ 1's batch (monitor slot q ^ 1)
                d = _lib.GenStatic.from_buffer_copy(self.fg.desc)
                d.noise_power_out = self.fg.npow[q ^ 1].data_ptr() if self.fg.npow is not None else None
                self.virt.append(d)
            # round 6: ... and the generator launch goes as well -- its workgroups ride on the step's bottleneck backward launch
            # (dccn_eq_buffers.gen_next_rides): the loop only ARMS the descriptor the step reads (labels, SNR row, batch offset)
            # (not under hipGraph replay of the step: the generator's per-batch arguments must stay outside a captured graph)
            self.ride_gen = bool(getattr(FLAGS, "generator_rides", True)) and not bool(getattr(FLAGS, "step_graph", False))
            self.pls[0].pipe_with(self.pls[1], 0, virt=self.virt[0], gen_rides=self.ride_gen)
            self.pls[1].pipe_with(self.pls[0], 1, virt=self.virt[1], gen_rides=self.ride_gen)
        self.nws = int(trainer.lib.dccn_eq_monitor_workspace_size(B, FLAGS.nsymbol, ofdmobj.K))
        self.ws = A.zeros(arena, self.nws, dtype=torch.uint8, device=dev)
        # round 6: in the pipelined loop the monitors are a job of the step's own optimizer launch (dccn_eq_buffers.monitor)
        self.mon = None
        if self.pipeline and getattr(FLAGS, "monitor_rides", True):
            from . import _lib as L_
            self.mon = []
            for q in range(2):
                plq = self.pls[q]
                npw = self.npow[q] if gen.want_noise_power else None
                self.mon.append(L_.EqMonitor(plq.chest.data_ptr(), self.H[q].data_ptr(), self.per_symbol, B, FLAGS.nsymbol, ofdmobj.K,
                                             plq.metrics_buf.data_ptr(), plq.tx_power.data_ptr(),
                                             None if npw is None else npw.data_ptr(), self.acc.data_ptr(), None,
                                             self.ws.data_ptr(), self.nws))
            kw = [dict(virt=self.virt[q], gen_rides=self.ride_gen) if self.virt is not None else {} for q in range(2)]
            self.pls[0].pipe_with(self.pls[1], 0, monitor=self.mon[0], **kw[0])
            self.pls[1].pipe_with(self.pls[0], 1, monitor=self.mon[1], **kw[1])
        self.i = 0
        # the step as 21 eager launches per call (default) or as a hipGraph replay: 0.188 vs 0.196 ms per loop step (tools/eqloop.py
        # --graph 0 / 1; host issue 0.156 vs 0.125 ms) -- the replay costs the GPU 3-7 us per step, the eager calls cost the host 30
        self.step_graph = bool(getattr(FLAGS, "step_graph", False))
        if self.overlap:
            self.side = torch.cuda.Stream(device=dev)
            self.ready = [torch.cuda.Event() for _ in range(2)]      # set q holds a finished batch
            self.done = [torch.cuda.Event() for _ in range(2)]       # the step that read set q has completed
            self.used = [False, False]
        self.pending = -1                                            # index (within the epoch) of the batch already generated
        self.fresh_epoch = True

    def _generate(self, i: int, q: int):
        from ._lib import check
        pl, gen = self.pls[q], self.gen
        if self.fg is not None:
            if self.virt is not None and i > 0:                      # frames only: the running step forms x in registers
                if getattr(self, "ride_gen", False):                 # ... and produces them: arm the descriptor ITS buffers hold
                    self.fg.arm(pl.bits, q, None, self.H[q], self.snr_rows[i], into=self.virt[q ^ 1])
                    return
                d = self.fg.arm(pl.bits, q, None, self.H[q], self.snr_rows[i])
                check(gen.lib.dccn_gen_static_frames(self.C.byref(d), gen._stream()), "dccn_gen_static_frames")
            else:
                self.fg.make_batch(pl.x, pl.bits, slot=q, out_H=self.H[q], snr=self.snr_rows[i])     # (advances gen.offset)
            return
        tx, _ = gen.transmit(pl.batch, out_bits=pl.bits)
        gen.channel(tx, self.snr_rows[i], out_x=pl.x, out_H=self.H[q], out_npow=self.npow[q])
        gen.offset += 1

    def begin_epoch(self, snr_table: np.ndarray):
        """snr_table [steps, B]: the epoch's per-frame training SNRs (one host draw + ONE copy per epoch, :407)"""
        torch = self.torch
        if self.overlap:
            torch.cuda.current_stream(self.tr.device).wait_stream(self.side)     # (nothing of the last epoch is in flight)
        self.snr.copy_(torch.from_numpy(np.ascontiguousarray(snr_table, dtype=np.float32).reshape(self.steps, -1)))
        self.acc.zero_()
        self.i = 0
        self.pending = -1
        self.fresh_epoch = True                                      # the side stream has not seen this epoch's SNR table yet

    def _produce(self, i: int, q: int):
        """batch i of the epoch into buffer set q, on the side stream"""
        torch = self.torch
        main = torch.cuda.current_stream(self.tr.device)
        if self.used[q]:
            self.side.wait_event(self.done[q])
        if self.fresh_epoch or not self.used[q]:
            self.side.wait_stream(main)                              # the SNR table copy / first use: the plan's construction
            self.fresh_epoch = False
        with torch.cuda.stream(self.side):
            self._generate(i, q)
            self.ready[q].record(self.side)
        self.pending = i

    def step(self):
        from ._lib import check
        tr = self.tr
        i = self.i % self.steps
        self.i += 1
        pipe = None
        if self.pipeline:
            q = i & 1
            if i == 0:
                self._generate(0, 0)
            if i + 1 < self.steps:
                self._generate(i + 1, q ^ 1)                         # before step i: its optimizer launch normalises it
            last = i + 1 == self.steps                               # the epoch's last step normalises nothing ahead
            pipe = (3 if last else 0) if i == 0 else (2 if last else 1)
        elif not self.overlap:
            q = 0
            self._generate(i, 0)
        else:
            q = i & 1
            if self.pending != i:                                    # first step of an epoch: nothing was produced ahead
                self._produce(i, q)
            if i + 1 < self.steps:
                self._produce(i + 1, q ^ 1)                          # the next batch, while this step runs
            self.torch.cuda.current_stream(tr.device).wait_event(self.ready[q])
        pl = self.pls[q]
        pl.run(True, pipe=pipe, graph=self.step_graph)
        if self.mon is not None and pipe is not None:
            return                                                   # (the step's optimizer launch carried the monitors)
        npow = self.npow[q] if self.gen.want_noise_power else None
        check(tr.lib.dccn_eq_monitor_accumulate(pl.chest.data_ptr(), self.H[q].data_ptr(), self.per_symbol, pl.batch,
                                                self.F.nsymbol, self.o.K, pl.metrics_buf.data_ptr(), pl.tx_power.data_ptr(),
                                                None if npow is None else npow.data_ptr(), self.acc.data_ptr(), None,
                                                self.ws.data_ptr(), self.nws, pl._stream()), "dccn_eq_monitor_accumulate")
        if self.overlap:
            self.done[q].record(self.torch.cuda.current_stream(tr.device))
            self.used[q] = True

    def epoch_means(self) -> np.ndarray:
        return self.acc.cpu().numpy() / max(self.steps, 1)


class BestSnapshot:
    """The best-train-loss model of :456-459, kept on the DEVICE: the reference calls ``saver.save`` every time the epoch loss
    improves; here that is a 28 MB device-to-device copy of the four arenas (microseconds) and the file is written when
    training ends (also when it ends with an exception or Ctrl-C: the epoch loop flushes in a ``finally``) -- and, checked
    once per epoch, every ``interval`` seconds in between, so a killed run loses at most that much plus one epoch.  Writing the
    checkpoint file on every improvement (60 small device-to-host copies + a 21 MB archive) cost as much as half an epoch
    of training steps."""
    NAMES = ("params", "adam_m", "adam_v", "adam_state")

    def __init__(self, trainer, path: str, FLAGS, interval: float = 60.0):
        import torch
        self.torch, self.tr, self.path, self.FLAGS, self.interval = torch, trainer, path, FLAGS, float(interval)
        self.bufs = {n: torch.empty_like(getattr(trainer, n)) for n in self.NAMES}
        self.dirty, self.written, self.t_last = False, False, time.time()

    def take(self):
        for n, b in self.bufs.items():
            b.copy_(getattr(self.tr, n))
        self.dirty = True
        self.maybe_flush()

    def maybe_flush(self):
        """call once per epoch, improved or not: a snapshot taken less than ``interval`` seconds after the last write is
        written by the first later epoch that finds the interval over"""
        if self.dirty and time.time() - self.t_last > self.interval:
            self.flush()

    def flush(self) -> str:
        """write the snapshot (not the trainer's current state) to ``path``; the trainer is left as it was"""
        if self.dirty:
            torch = self.torch
            with torch.no_grad():
                cur = {n: getattr(self.tr, n).clone() for n in self.NAMES}
                for n, b in self.bufs.items():
                    getattr(self.tr, n).copy_(b)
                save_checkpoint(self.path, self.tr, self.FLAGS)
                for n, c in cur.items():
                    getattr(self.tr, n).copy_(c)
            self.dirty, self.written, self.t_last = False, True, time.time()
        return self.path if self.written else ""


def device_epoch_runner(FLAGS, ofdmobj, trainer, gen, pl, steps: int = 197, overlap: bool = False,
                        pipeline: Optional[bool] = None):
    """tools/eqloop.py: one step of the loop below as a callable (SNR table drawn once)"""
    loop = DeviceEpochLoop(FLAGS, ofdmobj, trainer, gen, pl, steps, overlap=overlap, pipeline=pipeline)
    loop.begin_epoch(np.random.choice(TRAIN_SNR_GRID, [steps, pl.batch], p=TRAIN_SNR_PROB))
    return loop.step


def _train_on_device(FLAGS, ofdmobj, trainer, batch_size, frame_cnt, verbose, run_test):
    """the epoch loop of :func:`train` with every batch drawn on the GPU (static channels) straight into the fused
    plan's buffers; per-step scalars are accumulated on the device and fetched once per epoch."""
    import torch
    from .datagen import DeviceDataGen
    # :389-392,409: fading0 (static) unless --mobile, then the mixed-Doppler simulator (mobile=True, mix=True)
    gen = DeviceDataGen(FLAGS, ofdmobj, device=trainer.device, seed=FLAGS.seed, mobile=FLAGS.mobile, mix=FLAGS.mobile)
    pl, ev = trainer.resident(batch_size), trainer.resident(FLAGS.eval_frames)
    loss_min, epoch_min, best_path, history = 100.0, 0, "", []
    steps = frame_cnt // batch_size
    loop = DeviceEpochLoop(FLAGS, ofdmobj, trainer, gen, pl, steps, overlap=bool(getattr(FLAGS, "overlap_generator", False)),
                           pipeline=getattr(FLAGS, "pipeline_norm", None))
    best = BestSnapshot(trainer, os.path.join(FLAGS.save_dir, save_model_name(FLAGS)), FLAGS)
    try:
        for epoch in range(FLAGS.max_epoch_num):
            # (a RandomState of its own -- the same values as np.random.seed + np.random.choice -- so that chains training in
            # other threads of this process, config5.train_models, cannot interleave their draws with this one's)
            rs = np.random.RandomState((FLAGS.seed + 1000003 * (epoch + 1)) & 0xFFFFFFFF)
            # :407 one draw for the epoch's frames (the same stream of values as `steps` draws of one batch each)
            loop.begin_epoch(rs.choice(TRAIN_SNR_GRID, [steps, batch_size], p=TRAIN_SNR_PROB))
            for i in range(steps):
                loop.step()
            a = loop.epoch_means()
            train_loss_epoch = float(a[0])
            snr = rs.choice(TRAIN_SNR_GRID, [FLAGS.eval_frames], p=TRAIN_SNR_PROB)              # :438
            tx, _ = gen.transmit(FLAGS.eval_frames, out_bits=ev.bits)
            gen.channel(tx, snr, out_x=ev.x)
            gen.offset += 1
            ev.run(False)
            em = trainer._metrics(ev.metrics_buf, ev.tx_power)
            history.append(dict(epoch=epoch, train_loss=train_loss_epoch, train_ber=float(a[1]), chan_rms=float(a[4]),
                                test_loss=em["ce_mean"], test_ber=em["berlin"]))
            if verbose:
                print("Epoch: %d  Train Loss: %f  Tx Power: %f  Noise Power: %f  SNR MSE: %f | Test Loss: %f  Test BER: %.8f"
                      % (epoch, train_loss_epoch, a[2], a[3], a[4], em["ce_mean"], em["berlin"]))
            if train_loss_epoch < loss_min:
                epoch_min, loss_min = epoch, train_loss_epoch
                best.take()                                              # :456-459 (device snapshot; file written below)
            else:
                best.maybe_flush()                                       # an older snapshot whose interval has run out
            if epoch - FLAGS.early_stop > epoch_min:
                break
    finally:
        best_path = best.flush()                                         # also on exceptions / KeyboardInterrupt
    if verbose:
        print("Training Done!, Best model saved to\n%s" % best_path)
    result = dict(history=history, best_path=best_path, trainer=trainer)
    if run_test and best_path:
        load_checkpoint(best_path, trainer, with_optimizer=False)
        result["sweep"] = test_model_cross(FLAGS, trainer, ofdmobj, verbose=verbose)
    return result


def main(argv=None):
    FLAGS = parse_flags(argv)
    if FLAGS.test:                                                       # :243-254
        from .equalizer import EqualizerTrainer
        ofdmobj = ofdm.ofdm_tx(FLAGS)
        trainer = EqualizerTrainer(FLAGS, ofdmobj, load_rx_params(FLAGS), seed=FLAGS.seed)
        # the reference's test branch always looks for "_Equalizer<opt>_" (:251) although opt 0 is saved as
        # "_Equalizer_" (:333); accept either
        cands = [os.path.join(FLAGS.save_dir, FLAGS.token + "_Equalizer%d_" % FLAGS.opt + FLAGS.channel),
                 os.path.join(FLAGS.save_dir, save_model_name(FLAGS))]
        path = next((c for c in cands if os.path.exists(c + ".npz")), cands[0])
        load_checkpoint(path, trainer, with_optimizer=False)
        test_model_cross(FLAGS, trainer, ofdmobj)
        return
    t0 = time.time()
    train(FLAGS)
    print("wall time %.1f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
