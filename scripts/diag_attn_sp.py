"""which query tiles / key blocks of the hand-placed single-pass backward differ from the C++ form"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from check_attn_sp import make, bwd, rel
for (B, H, Nq, Nk) in [(1, 1, 640, 256), (1, 2, 640, 256), (1, 4, 640, 256), (2, 1, 640, 256)]:
    x = make(B, H, Nq, Nk, seed=1)
    dq2, dk2, dv2 = bwd(x, H, 2)
    dq3, dk3, dv3 = bwd(x, H, 3)
    print(f'Nq {Nq} Nk {Nk}: dq per 64-row tile', [f'{rel(dq2[:, t:t + 64], dq3[:, t:t + 64]):.1e}' for t in range(0, Nq, 64)],
          'dk per 64-key tile', [f'{rel(dk2[:, t:t + 64], dk3[:, t:t + 64]):.1e}' for t in range(0, Nk, 64)],
          'dv', [f'{rel(dv2[:, t:t + 64], dv3[:, t:t + 64]):.1e}' for t in range(0, Nk, 64)], flush=True)
