#!/usr/bin/env python3
"""Drop-in for the reference's train_iq.py: `from train_iq import TrainIQ` and the same CLI flags (train_iq.py:313-351).

    python train_iq.py --synthetic --hidden_dim 256 --latent_dim 256 --pwffn_dim 512 --num_layers 2 --total_training_steps 200
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bltvqg_amd  # noqa: E402,F401
from bltvqg_amd.train_iq import TrainIQ, SyntheticVocabulary, build_parser, main  # noqa: E402,F401

if __name__ == "__main__":
    sys.exit(main())
