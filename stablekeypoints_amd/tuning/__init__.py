"""GEMM algorithm selection for the frozen nn.Linear layers (hipBLASLt / rocBLAS through PyTorch's TunableOp).

`tunableop_gfx950.csv` holds the per-shape winners measured once on an MI355X for the GEMM shapes of the SD-1.5
512^2 step (`PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=<new file> python bench.py`, 2.5 minutes; new shapes only --
the entries of the new file are appended here) and, since round 2, of the T = 500, SD-2.1 768^2, SDXL 1024^2 and
fixed-global-batch bench configurations; since round 4 also of the augmented inference (10 views per image,
1 / 2 / 4 images per forward: `tools/infer_bench.py --dataset 4`, 73.5 -> 70.7 ms per image).  Round 5 added the 1- / 2-image
per-rank steps, the batched q | k | v projections of the self-attention blocks (one strided-batched GEMM with a broadcast A
operand + one [M, 3C] x [3C, C] input-gradient GEMM) and the batched context projections (one GEMM per layer width).  `enable()` loads it read-only: no
tuning happens at run time, shapes that are not in the file use the library default, and PyTorch ignores the
file when its validators (ROCm / hipBLASLt / rocBLAS versions, GPU architecture) do not match.  Results stay
fp32; measured -2.4 % step time (103.1 -> 100.7 ms).  `SKP_TUNABLEOP=0` turns it off."""
import os

_done = False


def enable() -> bool:
    global _done
    if _done or os.environ.get("SKP_TUNABLEOP", "1") == "0":
        return _done
    try:
        import torch
        from torch.cuda import tunable
        if not torch.cuda.is_available():
            return False
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")
        tunable.enable(True)
        tunable.tuning_enable(os.environ.get("PYTORCH_TUNABLEOP_TUNING", "0") == "1")
        if os.environ.get("PYTORCH_TUNABLEOP_TUNING", "0") != "1":
            # read-only use: nothing may be left behind at exit
            if hasattr(tunable, "write_file_on_exit"):
                tunable.write_file_on_exit(False)
            else:
                tunable.set_filename(os.devnull)        # this torch always dumps at exit: send it nowhere
        if os.path.exists(path):
            tunable.read_file(path)
        _done = True
    except Exception:                                   # older torch / no TunableOp: library defaults
        _done = False
    return _done
