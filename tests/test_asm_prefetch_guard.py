"""Build-time guard for the hand-issued prefetch of k_hist_rgb (image-analysis_amd/csrc/lfa_device.h, "vector memory operations
issued by hand"): the three `global_load_dwordx4 ... nt` of a round are written by an asm statement whose outputs the compiler
takes to be defined at once, while gfx9 does not interlock a VGPR with a load still in flight.  The loop is only correct if
nothing touches those registers between the request and the hand-counted `s_waitcnt vmcnt(N)` that follows it -- no copy, no
spill, no reuse.  Parity tests show that for today's compiler; this test shows it for whatever compiler builds the library:
it compiles k_threshold.hip to assembly with the library's own flags and checks every request of every k_hist_rgb variant.
CPU only (hipcc cross-compiles without a GPU)."""
import importlib.util
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _regs(text):
    """VGPR numbers named in an operand string: v7, v[10:13]"""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r"\bv(\d+)\b", text))
    return out


def _check(text):
    """Walk every path from every hand-issued request of every k_hist_rgb variant; returns the number of requests seen."""
    requests = 0
    for m in re.finditer(r"^(_ZN3lfa10k_hist_rgbILi\d+EEEvNS_3JobE):.*?^\.Lfunc_end", text, flags=re.S | re.M):
        name = m.group(1)
        lines = [ln.split(";")[0].strip() if "#ASM" not in ln else ln.strip() for ln in m.group(0).split("\n")]
        lines = [ln for ln in lines if ln and not ln.startswith((".", "#")) or ln.startswith(".LBB")]
        label = {ln[:-1]: i for i, ln in enumerate(lines) if ln.endswith(":")}
        for i, ln in enumerate(lines):
            if ln != ";;#ASMSTART" or not lines[i + 1].startswith("global_load_dwordx4"):
                continue
            loads = lines[i + 1:i + 4]
            assert all(t.startswith("global_load_dwordx4") for t in loads) and lines[i + 4] == ";;#ASMEND", (name, loads)
            dests = [frozenset(_regs(t.split(",")[0])) for t in loads]
            assert sum(len(d) for d in dests) == 12 and len(frozenset().union(*dests)) == 12
            requests += 1
            # every path from the request: nothing names a register of a load that a wait has not yet covered.
            # younger = vector memory operations issued on the path after the request (vmcnt is one in-order counter:
            # `vmcnt(n)` leaves at most n - younger of the three loads outstanding).
            todo, seen = [(i + 5, 3, 0)], set()
            while todo:
                k, pending, younger = todo.pop()
                while pending and (k, pending, min(younger, 4)) not in seen:
                    seen.add((k, pending, min(younger, 4)))
                    ins = lines[k]
                    k += 1
                    if ins.endswith(":") or ins.startswith(";;#"):
                        continue
                    w = re.match(r"s_waitcnt\b.*\bvmcnt\((\d+)\)", ins)
                    if w:
                        pending = min(pending, max(0, int(w.group(1)) - younger))
                        continue
                    live = frozenset().union(*dests[3 - pending:])
                    touched = _regs(ins) & live
                    assert not touched, f"{name}: `{ins}` names v{sorted(touched)} between the request and its wait"
                    assert not ins.startswith("scratch_"), f"{name}: spill between the request and its wait: `{ins}`"
                    if ins.startswith(("global_", "buffer_", "flat_")):
                        younger += 1
                    if ins.startswith("s_endpgm"):
                        break
                    assert not ins.startswith(("s_setpc", "s_swappc")), (name, ins)
                    br = re.match(r"s_(c?)branch(\w*)\s+(\S+)", ins)
                    if br:
                        t = label[br.group(3)]
                        skipped = 0
                        if br.group(2) == "_execz":
                            # The compiler's "no lane active" skip over the `if (g < ng)` block of k_hist_rgb: never taken,
                            # lane 0 of every round is inside the image (k_threshold.hip, comment at the plane store), and the
                            # hand-counted wait relies on exactly that -- the path is followed as if the store had been issued.
                            end = k                               # the skipped block ends at the label or at its jump to it
                            while end < len(lines) and lines[end] != br.group(3) + ":" and lines[end] != "s_branch " + br.group(3):
                                end += 1
                            skipped = sum(1 for x in lines[k:end] if x.startswith("global_store")) if end < len(lines) else 0
                        todo.append((t, pending, younger + skipped))
                        always = not br.group(1)
                        if br.group(2) == "_execnz":
                            # LLVM writes the uniform `break` of a loop as "s_cbranch_execnz <exit>" in a block of its own that
                            # never changes exec; a running wave has a lane, so that branch is always taken (the loops that do
                            # depend on exec clear lanes right before the branch)
                            b0 = k - 2
                            while b0 >= 0 and not lines[b0].endswith(":"):
                                b0 -= 1
                            always = not any(re.search(r"\bexec\b|^v_cmpx", x.split(",")[0]) for x in lines[b0 + 1:k - 1])
                        if always:
                            break
    return requests


@pytest.fixture(scope="module")
def hist_asm(tmp_path_factory):
    spec = importlib.util.spec_from_file_location("_lfa_build", os.path.join(ROOT, "image-analysis_amd", "_build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    asm = str(tmp_path_factory.mktemp("asm") / "k_threshold.s")
    flags = [f for f in b.FLAGS if f != "-fPIC"]
    subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", "-o", asm, os.path.join(b.CSRC, "k_threshold.hip")],
                   check=True, capture_output=True)
    return open(asm).read()


def test_nothing_touches_the_prefetched_registers_before_the_wait(hist_asm):
    assert _check(hist_asm) >= 12          # six rounding-order variants x (first request + in-loop request), at least


def test_the_guard_sees_a_planted_copy(hist_asm):
    """the same text with a copy of a requested register planted right after each in-loop request must be refused"""
    planted = re.sub(r"(global_load_dwordx4 v\[2:5\], v\[\d+:\d+\], off nt\n\s*;;#ASMEND\n)(?!\s*;;#ASMSTART\n\s*s_waitcnt)",
                     r"\1\tv_mov_b32_e32 v200, v3\n", hist_asm)
    assert planted != hist_asm
    with pytest.raises(AssertionError, match="between the request and its wait"):
        _check(planted)
