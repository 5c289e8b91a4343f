"""Basic-receiver harness: flags -> data -> fused GPU step -> checkpoint -> SNR sweep -> CSV.

Host-side mirror of dev/py/ofdmreceiver_np.py (H1-H4 of SURVEY.md section 8a):
  H1  the flag set and its defaults (:30-53)
  H2  the epoch data schedule and the BER-adaptive batch size (:211-243)
  H3  best-train-loss checkpointing and early stop (:268-274)
  H4  the final sweep SNR -10..30 dB x 20 000 frames -> ``Test_DCCN_<token>_<channel>.csv`` (:59-91)
The TF session.run calls become :class:`dl_ofdm_amd.engine.RxEngine` steps; data generation stays the
NumPy substrate (ofdm.py / radio.py), seeded instead of wall-clock seeded.

    python -m dl_ofdm_amd.receiver --channel=AWGN --nbits=2 --SNR=10 --nfilter=64 --max_epoch_num=20
"""
from __future__ import annotations

import argparse
import os
import time
from dataclasses import asdict, dataclass
from typing import Dict, Optional

import numpy as np

from . import _lib, ofdm, radio, sweep, util
from .engine import PARAM_NAMES, RxDims


def _bool(v):
    if isinstance(v, bool):
        return v
    return str(v).lower() in ("1", "true", "t", "yes", "y")


@dataclass
class Flags:
    """tf.app.flags of ofdmreceiver_np.py:30-53 (same names, same defaults)."""
    save_dir: str = "./output/"
    nbits: int = 1
    msg_length: int = 100800
    batch_size: int = 512
    max_epoch_num: int = 1000
    seed: int = 1
    nfft: int = 64
    nsymbol: int = 7
    npilot: int = 8
    nguard: int = 8
    nfilter: int = 80
    SNR: float = 3.0
    early_stop: int = 100
    ofdm: bool = True
    pilot: str = "lte"
    channel: str = "EPA"
    cp: bool = True
    longcp: bool = True
    load_model: bool = False
    split: float = 1.0
    token: str = "OFDM"
    test: bool = False
    # additions of this implementation (not in the reference)
    test_frames: int = 20000        # frames per sweep point (ofdmreceiver_np.py:69)
    eval_frames: int = 1024         # per-epoch evaluation batch (:249)
    snr_lo: int = -10
    snr_hi: int = 30                # inclusive (:72)
    device_data: bool = False       # generate bits/frames/channel/noise on the GPU (datagen.py) instead of NumPy
    align_window: bool = False      # extension: delay generated frames by the channel's centre-tap advance (datagen.py)
    tf_checkpoint: bool = False     # also write <save_dir>/<token>.index/.data-00000-of-00001 (tf.train.Saver format)
    iq_dump: bool = False           # per-epoch <token>_txiq.csv / _rxiq.csv constellation dumps (ofdmreceiver_np.py:264-265),
    #                                 written into save_dir; off by default: two syncs + two files per epoch


def parse_flags(argv=None) -> Flags:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for k, v in asdict(Flags()).items():
        t = _bool if isinstance(v, bool) else type(v)
        ap.add_argument("--" + k, type=t, default=v)
    return Flags(**vars(ap.parse_args(argv)))


def ideal_batch_size(berl_mean: float, nbits: int) -> int:
    """ofdmreceiver_np.py:242: frames per step grow as the BER falls (>= ~200 bit errors per step)."""
    return int(min(200.0 / max(berl_mean, 1.0e-6), 900000.0) / (55 * nbits)) // 8


def rx_dims(FLAGS: Flags, ofdmobj) -> RxDims:
    kin = (ofdmobj.K + ofdmobj.CP) if FLAGS.cp else ofdmobj.K
    return RxDims(S=FLAGS.nsymbol, kin=kin, F=FLAGS.nfilter, D=ofdmobj.frame_size, nbits=FLAGS.nbits)


def crop_cp(xs: np.ndarray, FLAGS: Flags, ofdmobj) -> np.ndarray:
    """The reference slices the cyclic prefix off inside the graph when cp=False (model.py:1236-1240);
    the normalisation is per position, so cropping before or after it is identical."""
    if FLAGS.cp:
        return xs
    return np.ascontiguousarray(xs[:, :, ofdmobj.CP:ofdmobj.CP + ofdmobj.K, :])


def make_batch(FLAGS: Flags, ofdmobj, fading, n_frames: int, snr_db):
    """bits -> OFDM frames -> fading -> AWGN  (ofdmreceiver_np.py:220-229, 73-79)."""
    ys = util.bit_source(FLAGS.nbits, ofdmobj.frame_size, n_frames)
    iq_cpx, _, _ = ofdmobj.ofdm_tx_frame_np(ys)
    xs, _ = fading.run(iq_cpx)
    snr = snr_db * np.ones((n_frames, 1)) if np.isscalar(snr_db) else snr_db
    xs, noise_pwr = radio.AWGN_channel_np(xs, snr)
    return crop_cp(xs.astype(np.float32), FLAGS, ofdmobj), ys.astype(np.int32), noise_pwr


# ---- checkpoints (reference variable names) ------------------------------
def save_checkpoint(path: str, eng, FLAGS: Flags):
    """<save_dir>/<token>.npz holding every variable tf.train.Saver would store: the 8 trainables, their
    Adam slots ``<var>/Adam`` / ``<var>/Adam_1``, ``global_step``, ``beta1_power``, ``beta2_power``."""
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    out: Dict[str, np.ndarray] = {}
    for n in PARAM_NAMES:
        out[n] = eng.view(n).detach().cpu().numpy()
        out[n + "/Adam"] = eng.view(n, eng.adam_m).detach().cpu().numpy()
        out[n + "/Adam_1"] = eng.view(n, eng.adam_v).detach().cpu().numpy()
    a = eng.adam()
    out["global_step"] = np.float32(a["global_step"])
    out["beta1_power"] = np.float32(a["beta1_power"])
    out["beta2_power"] = np.float32(a["beta2_power"])
    if getattr(FLAGS, "tf_checkpoint", False):                  # the reference's own format (tf.train.Saver, :191,271)
        from . import tf_bundle
        tf_bundle.write_checkpoint(path[:-4] if path.endswith(".npz") else path, tf_bundle.rx_to_tf(out, eng.dims.kin))
    out["__flags__"] = np.array(repr(asdict(FLAGS)))
    np.savez(path if path.endswith(".npz") else path + ".npz", **out)
    return path


class BestSnapshot:
    """The best-train-loss model of ofdmreceiver_np.py:268-272, kept on the DEVICE: the reference calls ``saver.save`` every time
    the epoch loss improves; here an improvement is a device-to-device copy of the four arenas and the file is written when
    training ends (also when it ends with an exception: the epoch loop flushes in a ``finally``) and, checked once per epoch,
    every ``interval`` seconds in between.  Writing the file on every improvement is 25 blocking device-to-host copies and a
    7 MB archive -- hundreds of times per run, a third of a BPSK receiver's training time, and (chains training in threads of
    one process, config5.train_models) time spent holding the interpreter lock.  The file holds the same bytes either way."""
    NAMES = ("params", "adam_m", "adam_v", "adam_state")

    def __init__(self, path: str, FLAGS, interval: float = 60.0):
        self.path, self.FLAGS, self.interval = path, FLAGS, float(interval)
        self.bufs, self.dirty, self.written, self.t_last = None, False, False, time.time()

    def take(self, eng):
        import torch
        if self.bufs is None:
            self.bufs = {n: torch.empty_like(getattr(eng, n)) for n in self.NAMES}
        for n, b in self.bufs.items():
            b.copy_(getattr(eng, n))
        self.dirty = True
        self.maybe_flush(eng)

    def maybe_flush(self, eng):
        if self.dirty and time.time() - self.t_last > self.interval:
            self.flush(eng)

    def flush(self, eng) -> str:
        """write the snapshot (not the engine's current state) to ``path``; the engine is left as it was"""
        if self.dirty:
            import torch
            with torch.no_grad():
                cur = {n: getattr(eng, n).clone() for n in self.NAMES}
                for n, b in self.bufs.items():
                    getattr(eng, n).copy_(b)
                save_checkpoint(self.path, eng, self.FLAGS)
                for n, c in cur.items():
                    getattr(eng, n).copy_(c)
            self.dirty, self.written, self.t_last = False, True, time.time()
        return self.path if self.written else ""


def read_checkpoint_file(path: str) -> Dict[str, np.ndarray]:
    """<path>.npz, or -- when only the TensorFlow bundle <path>.index/.data-* exists (a receiver trained by
    the reference) -- that, mapped to the engine's layouts."""
    stem = path[:-4] if path.endswith(".npz") else path
    if not os.path.exists(stem + ".npz") and os.path.exists(stem + ".index"):
        from . import tf_bundle
        return tf_bundle.rx_from_tf(tf_bundle.read_checkpoint(stem))
    z = np.load(stem + ".npz", allow_pickle=False)
    return {k: z[k] for k in z.files}


def load_checkpoint(path: str, eng, with_optimizer: bool = True):
    import torch
    z = read_checkpoint_file(path)
    eng.load_params({n: z[n] for n in PARAM_NAMES})
    if with_optimizer and eng.train and all((n + "/Adam") in z for n in PARAM_NAMES):
        for n in PARAM_NAMES:
            eng.view(n, eng.adam_m).copy_(torch.as_tensor(z[n + "/Adam"]))
            eng.view(n, eng.adam_v).copy_(torch.as_tensor(z[n + "/Adam_1"]))
        eng.adam_state.copy_(torch.tensor([float(z["global_step"]), float(z["beta1_power"]),
                                           float(z["beta2_power"]), 0.0]))


# ---- device-side data ----------------------------------------------------------------
def _device_gen(FLAGS: Flags, ofdmobj, device):
    from .datagen import DeviceDataGen
    return DeviceDataGen(FLAGS, ofdmobj, device=device, seed=FLAGS.seed)


def _gen_into(gen, eng, FLAGS: Flags, ofdmobj, n_frames: int, snr_db, slot: int = 0):
    """make_batch on the GPU, written into the engine's resident input buffers (cropped when cp=False); ``slot`` picks
    the label buffer (RxEngine.label_slot: the pipelined training loop fills one while a step reads the other)."""
    bits = eng.label_slot(slot) if slot else eng.bits
    if FLAGS.cp:
        return gen.make_batch(n_frames, snr_db, out_x=eng.x, out_bits=bits)[2]
    x, _, npow = gen.make_batch(n_frames, snr_db, out_bits=bits)
    eng.x.copy_(x[:, :, ofdmobj.CP:ofdmobj.CP + ofdmobj.K, :])
    return npow


# ---- sweep (H4) ------------------------------------------------------------------------------------
def test_model(FLAGS: Flags, model, ofdmobj=None, rank: int = 0, world: int = 1,
               device="cuda", out_dir: str = ".", verbose: bool = True, group=None):
    """SNR sweep of a trained receiver (ofdmreceiver_np.py:59-91), sharded over ``world`` ranks; rank 0
    writes ``Test_DCCN_<token>_<channel>.csv`` (columns SNR,BER,Loss).

    ``model``: the checkpoint path prefix, restored through the graph-name API exactly like the reference does
    (``load_model_np(path, session)`` then ``session.run([conf_matrix, berlin, ...], {x, y, SNR})``, :60,80), or an
    already loaded name -> array dict."""
    from .session import Session, load_model_np
    ofdmobj = ofdmobj or ofdm.ofdm_tx(FLAGS)
    snrs = list(range(FLAGS.snr_lo, FLAGS.snr_hi + 1))
    pts = sweep.make_points([FLAGS.nbits], [FLAGS.channel], snrs, base_seed=FLAGS.seed)
    sess = Session(device=device, seed=FLAGS.seed)
    if isinstance(model, str):
        y, x, _, _, _, _, berlin, conf_matrix, power_tx, noise_pwr, _, _, ce_mean, SNR = load_model_np(model, sess, FLAGS, ofdmobj)
    else:
        sess.restore(model, crop=None if FLAGS.cp else (ofdmobj.CP, ofdmobj.K), nsymbol=FLAGS.nsymbol)
        g = sess.get_tensor_by_name
        y, x, SNR = g("bits_in:0"), g("tx_ofdm:0"), g("SNR:0")
        berlin, conf_matrix, power_tx, noise_pwr, ce_mean = (g(n) for n in ("linear_ber:0", "conf_matrix:0", "tx_power:0",
                                                                             "noise_power:0", "ce_mean:0"))
    fading = radio.rayleigh_chan_lte(FLAGS, ofdmobj.Fs)
    gen = _device_gen(FLAGS, ofdmobj, device) if FLAGS.device_data else None
    if gen is not None:
        gen.want_noise_power = False          # the sweep table has no noise-power column

    def evaluate(p):
        if gen is not None:
            eng = sess.engine_for(FLAGS.test_frames, want_prob=False)
            gen.seed, gen.offset = p.seed, 0
            _gen_into(gen, eng, FLAGS, ofdmobj, FLAGS.test_frames, p.snr_db)
            eng.eval_step()
            m = eng.metrics()          # the float64 / int64 record itself: no fp32 ce_mean * count, no int32 counts
            if verbose:
                print("SNR: %.2f, BER: %.8f, Loss: %f" % (p.snr_db, m["berlin"], m["ce_mean"]))
            c = m["conf"]
            return [c[0][0], c[0][1], c[1][0], c[1][1], m["ce_sum"], m["count"]]
        else:
            np.random.seed(p.seed)
            test_ys = util.bit_source(FLAGS.nbits, ofdmobj.frame_size, FLAGS.test_frames)
            iq_cpx, _, _ = ofdmobj.ofdm_tx_frame_np(test_ys)
            test_xs, _ = fading.run(iq_cpx)
            snr_test = p.snr_db * np.ones((FLAGS.test_frames, 1))
            test_xs, _ = radio.AWGN_channel_np(test_xs, snr_test)
            confmax, berl, test_loss = sess.run([conf_matrix, berlin, ce_mean], {x: test_xs, y: test_ys, SNR: snr_test})
        if verbose:
            print("SNR: %.2f, BER: %.8f, Loss: %f" % (p.snr_db, berl, test_loss))
        count = float(confmax.sum())
        return [confmax[0][0], confmax[0][1], confmax[1][0], confmax[1][1], float(test_loss) * count, count]

    if gen is not None and not verbose:
        # nothing is printed per point: the whole shard + the final all-reduce stay on the stream (no host round trip)
        lib = _lib.load()

        def evaluate_into(p, row):
            eng = sess.engine_for(FLAGS.test_frames, want_prob=False)
            gen.seed, gen.offset = p.seed, 0
            _gen_into(gen, eng, FLAGS, ofdmobj, FLAGS.test_frames, p.snr_db)
            eng.eval_step()
            _lib.check(lib.dccn_metrics_table_add(eng.metrics_buf.data_ptr(), row.data_ptr(), eng._stream()), "table_add")
        table = sweep.run_sweep_device(pts, evaluate_into, rank, world, device=sess.device, group=group).cpu().numpy()
    else:
        table = sweep.run_sweep(pts, evaluate, rank, world, device=sess.device, group=group)
    sess.close()
    ber, loss = sweep.ber_loss(table)
    csvfile = os.path.join(out_dir, "Test_DCCN_%s.csv" % (FLAGS.token + "_" + FLAGS.channel))
    if rank == 0:
        sweep.write_csv(csvfile, snrs, ber, loss)
    return snrs, ber, loss, csvfile


# ---- training (H2, H3) ---------------------------------------------------------------------------
def train(FLAGS: Flags, device="cuda", verbose: bool = True, run_test: bool = True):
    from .engine import RxEngine
    ofdmobj = ofdm.ofdm_tx(FLAGS)
    dims = rx_dims(FLAGS, ofdmobj)
    frame_cnt = FLAGS.msg_length // FLAGS.nsymbol
    np.random.seed(FLAGS.seed)
    batch_size = FLAGS.batch_size // FLAGS.nsymbol                  # the flag counts OFDM symbols (:196)
    fading = radio.rayleigh_chan_lte(FLAGS, ofdmobj.Fs)
    snr_seq = np.zeros((8, 1), dtype=np.float32)                    # :209-210
    engines: Dict[int, RxEngine] = {}

    def engine_for(bs: int, src: Optional[RxEngine]) -> RxEngine:
        """The engine (arenas + plan) of the current batch size; optimizer state follows the training run.  The batch
        size only ever grows with the falling BER (:232-236), so the previous engine is released as soon as its state
        has been copied: one training engine is alive at a time."""
        if bs not in engines:
            engines[bs] = RxEngine(dims, bs, device=device, train=True, seed=FLAGS.seed, want_prob=False).pin_tuning()
        e = engines[bs]
        if src is not None and src is not e:
            e.params.copy_(src.params); e.adam_m.copy_(src.adam_m); e.adam_v.copy_(src.adam_v)
            e.adam_state.copy_(src.adam_state)
            for old_bs in [k for k, v in engines.items() if v is src]:
                engines.pop(old_bs).close_graph()
        return e

    eng = engine_for(batch_size, None)
    ev = RxEngine(dims, FLAGS.eval_frames, device=device, train=False, want_prob=False).pin_tuning()
    loss_min, epoch_min, best_path = 100.0, 0, ""
    history = []
    gen = _device_gen(FLAGS, ofdmobj, device) if FLAGS.device_data else None
    feed = None
    fused = {}                       # batch size -> datagen.FusedStaticGen
    import torch
    # device-generated runs keep the best model on the device and write it at the end (BestSnapshot); host-generated runs save
    # on every improvement as before
    best = BestSnapshot(os.path.join(FLAGS.save_dir, FLAGS.token), FLAGS) if gen is not None else None
    try:
        for epoch in range(FLAGS.max_epoch_num):
            np.random.seed(FLAGS.seed + 1000003 * (epoch + 1))           # reference: int(time.time()) + epoch
            train_snr = FLAGS.SNR + np.repeat(snr_seq, frame_cnt // 8, axis=0)
            n_use = train_snr.shape[0]
            losses, pwrs, berl = [], [], 0.5
            if gen is not None:
                # every step draws its own batch on the GPU into the engine's buffers; the per-step scalars the
                # reference fetches are accumulated on the device and read once per epoch (no per-step sync)
                mview = eng.metrics_buf.view(torch.float32)              # dccn_metrics: [12] ce_mean, [13] berlin
                acc = torch.zeros(3, dtype=torch.float32, device=eng.device)
                steps = n_use // batch_size
                # R0 is software-pipelined across the steps (RxEngine.train_step_pipelined): batch i+1 is generated into
                # eng.x / the other label slot before step i is issued, and normalised behind step i's Adam update
                # ... and the generator runs on its own stream (datagen.SideStreamFeeder): batch i+1 is produced while the forward
                # and backward launches of step i run; the step's last launch waits for it
                from .datagen import FusedStaticGen, SideStreamFeeder
                if FLAGS.cp and FusedStaticGen.supported(gen, eng) and not getattr(FLAGS, "no_fused_generator", False):
                    # round 5: static single-profile channels -- ONE C call per batch: the fused generator launch of the next batch
                    # + the four step launches, whose pipelined normalisation reads (y, noise, power partials) as its virtual
                    # input (include/dccn.h dccn_gen_static; datagen.FusedStaticGen).  Same batches as the loop below.
                    fg = fused.get(batch_size)
                    if fg is None:
                        fg = fused[batch_size] = FusedStaticGen(gen, batch_size, FLAGS.SNR, want_noise_power=True)
                    eng.drop_prefetch()
                    # (the three per-step monitors go onto the epoch accumulators in ONE stream-ordered library launch: three
                    # framework calls per step kept the interpreter lock for longer than the step's own C call, which is what
                    # several chains training in threads of one process -- config5.train_models -- then queue up behind)
                    mon = (eng.metrics_buf.data_ptr(), eng.tx_power.data_ptr(), (fg.npow[0].data_ptr(), fg.npow[1].data_ptr()),
                           acc.data_ptr())
                    for i in range(steps):
                        eng.train_step_generated(fg, slot=i & 1, last=(i + 1 == steps))
                        _lib.check(eng.lib.dccn_step_monitor_add(mon[0], mon[1], mon[2][i & 1], mon[3], eng._stream()),
                                   "dccn_step_monitor_add")
                    a = acc.cpu().numpy() / max(steps, 1)
                    losses, pwrs, noise_pwr = [float(a[0])], [float(a[1])], float(a[2])
                    berl = eng.metrics()["berlin"]
                    steps = 0                                            # (the loop below has nothing left to do)

                # the generator's noise-power monitor is reused per batch: each batch's value goes to one of two preallocated
                # slots (allocated on the main stream, never handed back to the caching allocator while a step may read them)
                noise_slots = torch.zeros(2, 1, dtype=torch.float32, device=eng.device)

                def make(slot, eng=eng, bs=batch_size):
                    npow = _gen_into(gen, eng, FLAGS, ofdmobj, bs, FLAGS.SNR, slot=slot)
                    if npow is None:
                        return None
                    noise_slots[slot].copy_(npow.reshape(1))
                    return noise_slots[slot]
                if feed is None or feed.eng is not eng:                  # one side stream + event pair per engine, not per epoch
                    feed = SideStreamFeeder(eng, make)
                else:
                    feed.rebind(make)
                noise_t = None
                if steps:
                    noise_t = make(0)
                    eng.prime()
                for i in range(steps):
                    last = i + 1 == steps
                    noise_next = None if last else feed.next((i + 1) & 1)
                    eng.train_step_pipelined(slot=i & 1, last=last, x_ready=None if last else feed.ready)
                    _lib.check(eng.lib.dccn_step_monitor_add(eng.metrics_buf.data_ptr(), eng.tx_power.data_ptr(),
                                                             None if noise_t is None else noise_t.data_ptr(), acc.data_ptr(),
                                                             eng._stream()), "dccn_step_monitor_add")
                    feed.step_issued()            # (after the monitor reads: the generator may now overwrite slot i & 1's values)
                    noise_t = noise_next
                if steps:
                    a = acc.cpu().numpy() / max(steps, 1)
                    losses, pwrs, noise_pwr = [float(a[0])], [float(a[1])], float(a[2])
                    berl = eng.metrics()["berlin"]
            else:
                xs, ys, noise_pwr = make_batch(FLAGS, ofdmobj, fading, n_use, train_snr)
                nb = n_use // batch_size
                for i in range(nb):
                    sl = slice(i * batch_size, (i + 1) * batch_size)
                    if i == 0:
                        eng.prime(xs[sl])
                    last = i + 1 == nb
                    eng.train_step_pipelined(next_x=None if last else xs[(i + 1) * batch_size:(i + 2) * batch_size],
                                             bits=ys[sl], last=last)
                    m = eng.metrics()
                    losses.append(m["ce_mean"]); pwrs.append(m["tx_power"]); berl = m["berlin"]
            train_loss_epoch = float(np.mean(losses))
            new_bs = max(batch_size, ideal_batch_size(berl, FLAGS.nbits))
            new_bs = min(new_bs, n_use)
            if new_bs != batch_size:
                eng = engine_for(new_bs, eng)
                batch_size = new_bs
            # per-epoch evaluation on fresh frames (:249-262)
            ev.params.copy_(eng.params)
            if gen is not None:
                _gen_into(gen, ev, FLAGS, ofdmobj, FLAGS.eval_frames, FLAGS.SNR)
                ev.eval_step()
            else:
                txs, tys, _ = make_batch(FLAGS, ofdmobj, fading, FLAGS.eval_frames, FLAGS.SNR)
                ev.eval_step(txs, tys)
            em = ev.metrics()
            if FLAGS.iq_dump:
                # constellation dumps of the graph's monitor branch (ofdmreceiver_np.py:256,264-265): first 2048 IQ pairs,
                # fp16.  The reference writes them into its working directory every epoch; here they go to save_dir, on request.
                from .session import monitor_tensors
                mon = monitor_tensors(ev, FLAGS.SNR * np.ones(FLAGS.eval_frames), FLAGS.seed, epoch + 1)
                os.makedirs(FLAGS.save_dir, exist_ok=True)
                np.savetxt(os.path.join(FLAGS.save_dir, "%s_txiq.csv" % FLAGS.token), mon["iq_tx"][:2048].cpu().numpy(), delimiter=",")
                np.savetxt(os.path.join(FLAGS.save_dir, "%s_rxiq.csv" % FLAGS.token), mon["iq_rx"][:2048].cpu().numpy(), delimiter=",")
            history.append(dict(epoch=epoch, train_loss=train_loss_epoch, test_loss=em["ce_mean"], test_ber=em["berlin"],
                                batch_size=batch_size))
            if verbose:
                print("Epoch: %d  Train Loss: %f  Tx Power: %f  Noise Power: %f | Test Loss: %f  Test BER: %.8f  next batch %d"
                      % (epoch, train_loss_epoch, float(np.mean(pwrs)), noise_pwr, em["ce_mean"], em["berlin"], batch_size))
            if train_loss_epoch < loss_min:                                # :268-272 (train loss selects the checkpoint)
                epoch_min, loss_min = epoch, train_loss_epoch
                if best is not None:
                    best.take(eng)                                         # (device snapshot; the file is written below)
                else:
                    best_path = save_checkpoint(os.path.join(FLAGS.save_dir, FLAGS.token), eng, FLAGS)
            elif best is not None:
                best.maybe_flush(eng)
            if epoch - FLAGS.early_stop > epoch_min:
                break
    finally:
        if best is not None:
            best_path = best.flush(eng)                                # also on exceptions / KeyboardInterrupt
    if verbose:
        print("Training Done!, Best model saved to\n%s" % best_path)
    result = dict(history=history, best_path=best_path, params=eng.get_params())
    if run_test and best_path:
        result["sweep"] = test_model(FLAGS, best_path, ofdmobj, device=device, verbose=verbose)
    return result


def main(argv=None):
    FLAGS = parse_flags(argv)
    if FLAGS.test:
        test_model(FLAGS, os.path.join(FLAGS.save_dir, FLAGS.token))
        return
    t0 = time.time()
    train(FLAGS)
    print("wall time %.1f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
