"""Build-time checks on the compiled code objects that need no GPU (hipcc cross-compiles gfx950 here)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = pytest.mark.skipif(not (os.path.exists(hipcc) or shutil.which("hipcc")), reason="hipcc not available")


def test_inline_asm_transposing_reads_are_waited_for():
    """csrc/wgrad_bf16_dma.hip issues its `ds_read_b64_tr_b16` fragment reads as inline asm with hand-placed COUNTED
    `s_waitcnt lgkmcnt(N)`; the compiler does not know the destinations are written asynchronously.  Checked on the ISA of every
    instantiation: no instruction touches a read's destination before a wait has retired it, and no scalar-memory load (out-of-order
    return) is in flight beside the reads."""
    import check_asm_waits as C
    seen, total, errs = C.check_file(os.path.join(ROOT, "pangu-pytorch_amd", "csrc", "wgrad_bf16_dma.hip"))
    assert seen >= 4 and total >= 60, (seen, total)
    assert not errs, errs[:5]


def test_wait_checker_sees_violations():
    """The checker itself: an early use, an under-counted wait and an SMEM load beside the reads are all reported."""
    import check_asm_waits as C
    ok = ["ds_read_b64_tr_b16 v[4:5], v1 offset:0", "ds_read_b64_tr_b16 v[6:7], v1 offset:64", "s_waitcnt lgkmcnt(1)",
          "v_mfma_f32_16x16x32_bf16 v[20:23], v[4:5], v[8:9], v[20:23]", "s_waitcnt lgkmcnt(0)", "v_mov_b32 v30, v6"]
    assert C.check_kernel("ok", ok)[1] == []
    early = ["ds_read_b64_tr_b16 v[4:5], v1 offset:0", "v_mov_b32 v30, v4", "s_waitcnt lgkmcnt(0)"]
    assert len(C.check_kernel("early", early)[1]) == 1
    under = ["ds_read_b64_tr_b16 v[4:5], v1 offset:0", "ds_read_b64_tr_b16 v[6:7], v1 offset:64", "s_waitcnt lgkmcnt(1)",
             "v_mov_b32 v30, v7"]
    assert len(C.check_kernel("under", under)[1]) == 1
    smem = ["ds_read_b64_tr_b16 v[4:5], v1 offset:0", "s_load_dwordx2 s[2:3], s[0:1], 0x0", "s_waitcnt lgkmcnt(0)"]
    assert len(C.check_kernel("smem", smem)[1]) == 1
