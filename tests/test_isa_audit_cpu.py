"""Static check of the compiled kernels (no GPU): every LDS-DMA hand-over barrier is preceded, in the issuing wave, by
`s_waitcnt vmcnt(0)`.  Round 3 found that hipcc inserts that wait for `__syncthreads()` behind LDS-DMA builtins in the
stand-alone GEMM kernels but NOT in the persistent ones (the consumers could read the previous occupant of the X-tile
buffer whenever the tile arrived later than their own first weight fragments - seen as rare one-segment errors with
cross-XCD hand-offs); the waits are explicit now and this audit keeps them there."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_lds_dma_handover_barrier_waits_for_the_dma():
    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "tools", "isa_audit.py"))
    audit = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(audit)
    txt = audit.compile_asm([])
    seen, bad = 0, []
    for name, lines in audit.kernels(txt):
        n, findings = audit.audit(name, lines)
        seen += n
        if findings:
            bad.append((name, sorted({b for _, b in findings})))
    assert seen >= 100, seen          # the GEMM flavours, the two persistent kernels: > 100 LDS-DMA sites
    assert not bad, bad
