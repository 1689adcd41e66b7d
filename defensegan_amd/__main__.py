"""``python -m defensegan_amd``: project a stack of images onto a generator's range from the command line, with the
reference's reconstruction flags (--cfg, --rec_iters, --rec_rr, --rec_lr, --batch_size, --override, --same_init,
--rec_path: whitebox.py:358-395 / blackbox.py:714-762) plus the generator weights and the input/output arrays.

    python -m defensegan_amd --cfg mnist --init_path output/gans/mnist --input x_adv.npy --output rec.npy --rec_rr 10

``--input`` is an .npy of images [n,H,W,C] already in generator range ([0,1] MNIST/F-MNIST, [-1,1] CelebA) unless
``--raw`` (then [0,255] and the reference's input transform is applied, gan.py:684-685, 764-765).  One rank per GPU under
torchrun shards the images contiguously and rank 0 writes the gathered result."""
import argparse
import os
import sys

import numpy as np

from . import config, gan_defense
from .gan import gan_from_config


def build_parser() -> argparse.ArgumentParser:
    ap = config.add_rec_flags(argparse.ArgumentParser(prog="python -m defensegan_amd", description=__doc__.split("\n\n")[0]))
    ap.add_argument("--init_path", required=True, help="generator weights: TensorFlow checkpoint dir/prefix or .npz pack")
    ap.add_argument("--input", required=True, help=".npy of images [n,H,W,C]")
    ap.add_argument("--output", required=True, help=".npy to write the reconstructions to")
    ap.add_argument("--raw", action="store_true", help="input is [0,255]: apply the reference's input transform first")
    ap.add_argument("--seed", type=int, default=11241990, help="seed of the z0 draw (whitebox.py:143)")
    ap.add_argument("--no_coalesce", action="store_true", help="one engine call per --batch_size batch, as the reference's loop (slower, same results)")
    return ap


def resolve_cfg(spec: str) -> str:
    return spec if (os.path.exists(spec.split("#", 1)[0])) else config.builtin_cfg(spec)


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)
    cfg_path = resolve_cfg(args.cfg)
    cfg = config.load_config(cfg_path)
    rp = config.resolve_rec_params(cfg, args)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        use_cuda = torch.cuda.is_available()
        if use_cuda:
            torch.cuda.set_device(local)         # before any CUDA / RCCL use: one process per GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if use_cuda else "gloo",
                                **({"device_id": torch.device("cuda", local)} if use_cuda else {}))
    gan = gan_from_config(cfg_path, rec_rr=rp["rec_rr"], rec_iters=rp["rec_iters"], rec_lr=rp["rec_lr"], device=local)
    gan.load_generator(args.init_path)
    x = np.load(args.input).astype(np.float32)
    if args.raw:
        x = gan.input_transform(x)
    x = x.reshape([-1] + list(gan.image_dim))
    n = len(x)
    s, e = gan_defense.shard_range(n, rank, world)
    z_same = None
    if args.same_init:                                       # whitebox.py:181-183: one sigma = 1 draw reused for every batch
        z_same = np.random.RandomState(args.seed).randn(rp["batch_size"] * rp["rec_rr"], int(gan.latent_dim)).astype(np.float32)
    # the reference's loop feeds --batch_size images per session.run (blackbox.py:537-541); here runs of whole batches go to the
    # engine in one call (same latents per batch, --same_init restarting at every batch: the same bits -- tests/test_gpu_coalesce.py --
    # at 0.85 instead of 0.67 of the peak for the default batch of 50); with USE_BN a batch is the unit of the statistics: one call each
    out = gan_defense.project_in_batches(gan.reconstruct, x[s:e], rp["batch_size"], rp["rec_rr"], seed=args.seed, first_image=s,
                                         same_init_z=z_same, coalesce=False if args.no_coalesce else None)
    if world > 1:
        out = gan_defense.gather_shards(out, n)          # one tensor all_gather (RCCL over xGMI), no pickling
        dist.destroy_process_group()
    if rank == 0:
        np.save(args.output, out)
        print("projected %d images (R=%d, L=%d, lr=%g) -> %s" % (n, rp["rec_rr"], rp["rec_iters"], rp["rec_lr"], args.output))
    return 0


if __name__ == "__main__":
    sys.exit(main())
