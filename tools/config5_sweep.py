"""CLI of dl_ofdm_amd/config5.py (BASELINE.json config[4]); shards over the ranks of a torch.distributed job.

    python tools/config5_sweep.py --out profiles/r02_config5 [--frames 20000] [--eq_epochs 600] [--backend nccl|gloo]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/config5_sweep.py --out ...
    python tools/config5_sweep.py --share 4 --out ...     # ONE GPU: four ranks (gloo) share it, one hardware queue each
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_ofdm_amd import config5     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="config5_out")
    ap.add_argument("--frames", type=int, default=20000)
    ap.add_argument("--eq_epochs", type=int, default=600, help="<= 0: the reference's 4000 * nbits cap")
    ap.add_argument("--rx_epoch_scale", type=float, default=1.0)
    ap.add_argument("--classical_frames", type=int, default=1500)
    ap.add_argument("--ckpt_dir", default="", help="where the best-model checkpoints go (default: a temporary directory; they are "
                                                   "~90 MB per rank and would push a gpurun_out/ result directory over its 64 MiB limit)")
    ap.add_argument("--backend", default=None, help="nccl (default; RCCL over xGMI) or gloo; also DCCN_DIST_BACKEND")
    ap.add_argument("--nbits", default="1,2,3,4", help="modulations (bits per symbol), comma separated")
    ap.add_argument("--channels", default=",".join(config5.CHANNELS))
    ap.add_argument("--snrs", default="", help="comma separated SNRs in dB (default: -10..29)")
    ap.add_argument("--classical_every", type=int, default=3)
    ap.add_argument("--share", type=int, default=0, help="started as ONE process: re-launch as this many gloo ranks sharing the "
                                                          "visible GPU(s) (the four training chains run side by side)")
    ap.add_argument("--chain_streams", type=int, default=-1,
                    help="training chains a rank runs next to each other, each in a thread with a HIP stream of its own "
                         "(default: all the chains the rank owns; 1: one after the other -- same results)")
    ap.add_argument("--chain_group", type=int, default=-1,
                    help="this many of a rank's chains (those with the smallest epoch budgets) train their equalisers as ONE chain "
                         "group on one stream (default: 2 of >= 4 chains; 0: none -- same results)")
    a = ap.parse_args()
    if a.share > 1 and "RANK" not in os.environ:
        import subprocess
        argv, skip = [], False
        for t in sys.argv[1:]:
            if skip:
                skip = False
            elif t == "--share":
                skip = True
            elif not t.startswith("--share="):
                argv.append(t)
        if not any(t == "--backend" or t.startswith("--backend=") for t in argv):
            argv += ["--backend", "gloo"]
        raise SystemExit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nproc-per-node", str(a.share),
                                          "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29517"),
                                          os.path.abspath(__file__)] + argv))
    import torch
    rank, world, local = config5.init_distributed(a.backend)
    import tempfile
    snrs = tuple(int(v) for v in a.snrs.split(",")) if a.snrs else config5.SNRS
    config5.run(a.out, a.frames, a.eq_epochs, a.classical_frames, a.rx_epoch_scale,
                nbits_list=tuple(int(v) for v in a.nbits.split(",")), channels=tuple(a.channels.split(",")), snrs=snrs,
                classical_every=a.classical_every, rank=rank, world=world,
                device="cuda:%d" % local, ckpt_dir=a.ckpt_dir or tempfile.mkdtemp(prefix="dccn_c5_"),
                chain_streams=None if a.chain_streams < 0 else a.chain_streams,
                chain_group=None if a.chain_group < 0 else a.chain_group)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
