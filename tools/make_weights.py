#!/usr/bin/env python3
"""Write seeded synthetic weight packs (.sdw) for the two networks.

The reference's real model files (pipeline/model/segment2.onnx, emd4.onnx) are missing
from the checkout (.MISSING_LARGE_BLOBS); there is no network to fetch checkpoints.  These
packs have the published architectures' exact tensor shapes with N(0, 1/fan_in) weights,
so throughput is representative and HIP-vs-oracle parity is exact-shape.

usage: make_weights.py OUT_DIR [--seed-seg N] [--seed-emb N]  -> OUT_DIR/segment.sdw, embedding.sdw
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pyannote-audio_speaker-diarization_cpp_amd"))
import weightpack  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir")
    ap.add_argument("--seed-seg", type=int, default=4321)
    ap.add_argument("--seed-emb", type=int, default=4322)
    a = ap.parse_args()
    os.makedirs(a.out_dir, exist_ok=True)
    weightpack.save_pack(os.path.join(a.out_dir, "segment.sdw"), weightpack.synth_segmentation_weights(a.seed_seg))
    weightpack.save_pack(os.path.join(a.out_dir, "embedding.sdw"), weightpack.synth_embedding_weights(a.seed_emb))
    print(os.path.join(a.out_dir, "segment.sdw"), os.path.join(a.out_dir, "embedding.sdw"))


if __name__ == "__main__":
    main()
