#!/usr/bin/env python3
"""Rehearsal of the multi-rank solve on a ONE-GPU box: N processes (gloo for the exchange), every one with its own
engine on GPU 0.  Exercises the real engines together with the DP all-gather and the sharded table.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 tools/rehearse_multi.py [bits] [dp_bits] [gx gy]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd import GPUEngine, dist, host, solve

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 56
dp = int(sys.argv[2]) if len(sys.argv) > 2 else 10
gx, gy = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (64, 128)
rank, _, world = dist.init(backend="gloo")
priv = (0xD1B54A32D192ED03C0FFEE1234567890ABCDEF % (1 << (bits - 1))) | (1 << (bits - 1))
pub = host.ec_mul_g(priv)
key, st = solve.solve_key_multi(0, (1 << bits) - 1, pub, dp_bits=dp, seed=3, max_launches=20000,
                                engine_factory=lambda lr: GPUEngine(gx, gy, 0, 1 << 17))
print(f"rank {rank}/{world}: key {'FOUND' if key == priv else 'WRONG'} {key:#x}  launches {st['launches']} "
      f"DPs {st['dps']} table(this rank) {st['table_this_rank']} jumps 2^{__import__('math').log2(st['jumps']):.2f} "
      f"{st['jumps'] / st['seconds'] / 1e9:.2f} G jumps/s", flush=True)
sys.exit(0 if key == priv else 1)
