"""Raw PRG launch (fss_amd_prg_gen) timings: mul 1/2/4, AES and ChaCha, 2^20 and 2^24 seeds.  HIP events on the
current stream; prints ms, G blocks/s and the LDS fraction for AES (160 lookups x 4 B per block, 78.6 TB/s)."""
import sys
import torch
from fss_amd.engine import Engine

LDS_PEAK = 256 * 128 * 2.4e9


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    for prg in ("aes128_mmo", "chacha"):
        eng = Engine(prg=prg)
        for lg in (20, 24):
            n = 1 << lg
            seeds = torch.randint(-2**31, 2**31, (n, 4), dtype=torch.int32, device="cuda")
            for mul in (1, 2, 4):
                for _ in range(5):
                    eng.prg_gen(mul, seeds)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    eng.prg_gen(mul, seeds)
                b.record()
                torch.cuda.synchronize()
                ms = a.elapsed_time(b) / reps
                blocks = n * mul / (ms * 1e-3)
                frac = blocks * 640 / LDS_PEAK if prg == "aes128_mmo" else float("nan")
                print(f"{prg:10s} 2^{lg} mul={mul}  {ms:8.4f} ms  {blocks / 1e9:7.2f} G blocks/s  lds_frac {frac:.3f}  "
                      f"hbm {n * 16 * (1 + mul) / (ms * 1e-3) / 1e12:.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
