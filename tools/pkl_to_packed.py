#!/usr/bin/env python3
"""Convert the reference's dataset cache to the packed layout and back.

The reference stores a dataset as `processed_data.pkl` beside the SMILES file: a torch.save'd
list of process_step1 dicts (src/data/bpe_dataset.py:23-26, 47-54, 170).  The GPU collation
reads the packed int32 arrays of ps_vae_amd/data/packed.py instead (an .npz file).

    python tools/pkl_to_packed.py data/zinc250k/processed_data.pkl data/zinc250k/train.npz
    python tools/pkl_to_packed.py --to-pkl data/zinc250k/train.npz out/processed_data.pkl

The pickle is read with torch.load(weights_only=True): a file that holds anything but plain
lists / dicts / numbers / tensors is refused rather than executed.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main(argv=None):
    import torch

    import ps_vae_amd  # noqa: F401
    from ps_vae_amd.data.packed import PackedMolecules, load_step1_pickle
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('src')
    ap.add_argument('dst')
    ap.add_argument('--to-pkl', action='store_true', help='packed .npz -> list of process_step1 dicts')
    args = ap.parse_args(argv)
    if args.to_pkl:
        packed = PackedMolecules.load(args.src)
        torch.save([packed.to_step1(i) for i in range(len(packed))], args.dst)
        print(f'{len(packed)} molecules -> {args.dst}')
    else:
        packed = PackedMolecules.from_step1(load_step1_pickle(args.src))
        packed.save(args.dst)
        print(f'{len(packed)} molecules, {int(packed.n_atoms.sum())} atoms -> {args.dst}')


if __name__ == '__main__':
    main()
