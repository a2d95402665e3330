#!/usr/bin/env python
"""`python train.py ...` as run.sh:109-140 calls it (from the repository root): mtn_amd.train with the same flags."""
from mtn_amd.train import main

if __name__ == "__main__":
    main()
