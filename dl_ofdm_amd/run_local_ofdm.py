"""Sweep driver -- the in-process counterpart of dev/py/run_local_ofdm.py:
  stage 1 (:66-90)  one basic-receiver training + BER sweep per (longcp, modulation, cp) on AWGN;
  stage 2 (:92-118) the channel equaliser on top of the BPSK receivers: --opt=0, channel mixRayleigh, cp in
                    (True, False), 4000*nbits epochs, cross-channel test CSVs.
The reference spawns one OS process per configuration and runs them one after the other; here the configurations
are independent units dealt round-robin to the ranks of a ``torch.distributed`` job (one process per GPU), and a
configuration whose result CSV already exists is skipped, as in the reference (:82-86, :108-112).

    python -m dl_ofdm_amd.run_local_ofdm --awgn=True [--equalizer=True] [--device_data=True] [--max_epoch_scale 0.01]
    python -m torch.distributed.run --nproc-per-node 8 -m dl_ofdm_amd.run_local_ofdm
"""
from __future__ import annotations

import argparse
import os

from .receiver import Flags, _bool, train


def configurations(nfft: int = 64, batchsize: int = 512, ebno: float = 5.0, epoch_scale: float = 1.0):
    """The driver's grid: longcp in (False, True) x nbits 4..1 x cp in (False, True), channel AWGN,
    SNR = 5 dB per bit, 1200*nbits epochs, early stop 200, nfilter = nfft (run_local_ofdm.py:61-80)."""
    token = "OFDM_Dense3"
    out = []
    for longcp in (False, True):
        save_dir = "./ofdm_lte_ext_%d_%scp_mobile/" % (nfft, "long" if longcp else "short")
        result_dir = "./test_ext_%d_%s_cross_mobile" % (nfft, "long" if longcp else "short")
        for nbits in (4, 3, 2, 1):
            snr = float(ebno * nbits)
            for cp in (False, True):
                token1 = "%s_%dmod_snr%d_cp%s" % (token, nbits, int(snr), cp)
                out.append((Flags(channel="AWGN", save_dir=save_dir, early_stop=200, nfilter=nfft, batch_size=batchsize,
                                  max_epoch_num=max(1, int(1200 * nbits * epoch_scale)), cp=cp, nfft=nfft, longcp=longcp,
                                  SNR=snr, nbits=nbits, token=token1), result_dir))
    return out


def equalizer_configurations(nfft: int = 64, batchsize: int = 512, ebno: float = 5.0, epoch_scale: float = 1.0,
                             learning: float = 0.001, mobile: bool = False):
    """run_local_ofdm.py:92-118: nbits = 1, opt = 0, channel mixRayleigh, cp in (True, False), per longcp."""
    from .receiver_mp import Flags as EqFlags
    token, nbits, opt = "OFDM_Dense3", 1, 0
    out = []
    for longcp in (False, True):
        save_dir = "./ofdm_lte_ext_%d_%scp_mobile/" % (nfft, "long" if longcp else "short")
        result_dir = "./test_ext_%d_%s_cross_mobile" % (nfft, "long" if longcp else "short")
        snr = float(ebno * nbits)
        for cp in (True, False):
            token1 = "%s_%dmod_snr%d_cp%s" % (token, nbits, int(snr), cp)
            out.append((EqFlags(channel="mixRayleigh", save_dir=save_dir, init_learning=learning, early_stop=200,
                                nfilter=nfft, batch_size=batchsize, max_epoch_num=max(1, int(4000 * nbits * epoch_scale)),
                                cp=cp, nfft=nfft, longcp=longcp, opt=opt, mobile=mobile, SNR=snr, nbits=nbits,
                                token=token1), result_dir))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--awgn", type=_bool, default=True)
    ap.add_argument("--equalizer", type=_bool, default=False, help="also run stage 2 (needs stage 1's checkpoints)")
    ap.add_argument("--mobile", type=_bool, default=False)
    ap.add_argument("--device_data", type=_bool, default=False, help="draw every batch on the GPU (datagen.py)")
    ap.add_argument("--max_epoch_scale", type=float, default=1.0, help="scale the reference's 1200*nbits epochs")
    ap.add_argument("--msg_length", type=int, default=100800)
    ap.add_argument("--test_frames", type=int, default=20000)
    args = ap.parse_args(argv)
    import torch
    from .config5 import init_distributed
    # one process per GPU; the process group (backend nccl = RCCL, or DCCN_DIST_BACKEND=gloo for control-flow tests on
    # a box with fewer GPUs than ranks) is only used as a barrier between the stages: every configuration is trained
    # and swept by ONE rank (sweeps inside run with world = 1 and never reduce across ranks)
    rank, world, local = init_distributed()
    if args.awgn:
        for i, (flags, result_dir) in enumerate(configurations(epoch_scale=args.max_epoch_scale)):
            if i % world != rank:
                continue
            os.makedirs(result_dir, exist_ok=True)
            csvdest = os.path.join(result_dir, "Test_DCCN_%s_%s.csv" % (flags.token, flags.channel))
            if os.path.isfile(csvdest):
                continue
            flags.msg_length, flags.test_frames, flags.device_data = args.msg_length, args.test_frames, args.device_data
            res = train(flags, device="cuda:%d" % local)
            if "sweep" in res and os.path.isfile(res["sweep"][3]):
                os.replace(res["sweep"][3], csvdest)
    if args.equalizer:
        if world > 1:
            torch.distributed.barrier()                               # stage 2 reads stage 1's checkpoints
        from . import receiver_mp
        for i, (flags, result_dir) in enumerate(equalizer_configurations(epoch_scale=args.max_epoch_scale,
                                                                         mobile=args.mobile)):
            if i % world != rank:
                continue
            os.makedirs(result_dir, exist_ok=True)
            last = "Test_DCCN_%s_Equalizer%d_%s_test_chan_Custom%s.csv" % (flags.token, flags.opt, flags.channel,
                                                                          "_mobile" if flags.mobile else "")
            if os.path.isfile(os.path.join(result_dir, last)):
                continue
            flags.msg_length, flags.device_data = args.msg_length, args.device_data
            res = receiver_mp.train(flags, device="cuda:%d" % local)
            for ch, (_, _, _, path) in res.get("sweep", {}).items():
                if os.path.isfile(path):
                    os.replace(path, os.path.join(result_dir, os.path.basename(path)))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
