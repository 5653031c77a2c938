#!/usr/bin/env python3
"""Noise-only inference driver in the shape of the reference's infer.py (a batch of 8 Gaussian rolls through
the reverse chain).  The reference's infer.py targets its U-Net prototype, which cannot be constructed at the
reference commit; this driver runs the same noise-only flow on ClassifierFreeDiffRoll."""
import sys

from diffroll_amd.cli import main

if __name__ == "__main__":
    main(["task=generation", "dataset=Sampling", "dataset.num_samples=8", "dataloader.batch_size=8"] + sys.argv[1:])
