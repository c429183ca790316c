"""Static check of the hand-scheduled pipelines (dyn_phase_kernel, rom_phase_kernel): their prefetch loads are issued
as asm into AGPRs behind the compiler's back and waited for with explicit counted s_waitcnt (kernels.hip, `aload`).  The
protocol is only sound if NO instruction touches such an AGPR between its load and the wait that covers it -- the
compiler does not know the register is in flight, so a live-range split or an early copy would read stale data.  This
test disassembles the device code and replays every instantiation's instruction stream against an in-order model of the
vector-memory queue (gfx9: loads and stores retire through one counter, in order)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "towr_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

AGPR = re.compile(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b")


def _agprs(text):
    out = set()
    for m in AGPR.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def _vmcnt(ins):
    m = re.match(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)$", ins)
    if m:   # raw immediate (the asm waits): vmcnt = bits 3:0 and 15:14
        imm = int(m.group(1), 0)
        return (imm & 0xF) | (((imm >> 14) & 3) << 4)
    m = re.search(r"vmcnt\((\d+)\)", ins)
    return int(m.group(1)) if m and ins.startswith("s_waitcnt") else None


def _functions(asm):
    cur, name = None, None
    for line in asm.splitlines():
        m = re.match(r"^(_ZN3twr\w+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                yield name, cur
                cur = None


def _replay(lines):
    """Returns (violations, n_async_loads).  The loop blocks are replayed twice (what is in flight at the end of an
    iteration meets the blocks the next iteration starts with)."""
    idx = [i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l) and ("Loop Header" in l or "in Loop:" in l)]
    if not idx:
        order = list(range(len(lines)))
    else:
        start = idx[0]
        end = len(lines)
        for i in range(idx[-1] + 1, len(lines)):      # first block label after the last in-loop label = the exit block
            if re.match(r"^\.LBB\d+_\d+:", lines[i]):
                end = i
                break
        order = list(range(0, end)) + list(range(start, end)) + list(range(end, len(lines)))
    queue, flying, bad, n_loads = [], set(), [], 0
    for i in order:
        ins = lines[i].split(";")[0].strip()
        if not ins or ins.endswith(":") or ins.startswith("."):
            continue
        n = _vmcnt(ins)
        if n is not None:
            keep = queue[len(queue) - n:] if n else []
            queue = keep
            flying = set().union(*[q for q in queue]) if queue else set()
            continue
        regs = _agprs(ins)
        if re.match(r"(global|scratch|buffer|flat)_load", ins):
            dst = _agprs(ins.split(",")[0])
            if dst & flying:
                bad.append((i, ins, "loads into registers that are still in flight"))
            queue.append(dst)          # (a load into VGPRs is an entry without registers)
            flying |= dst
            n_loads += bool(dst)
            continue
        if regs & flying:
            bad.append((i, ins, "touches a[%s] before the wait that covers the load" % ",".join(map(str, sorted(regs & flying)))))
        if re.match(r"(global|scratch|buffer|flat)_(store|atomic)", ins):
            queue.append(set())
    return bad, n_loads


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    if not shutil.which(HIPCC):
        pytest.skip("hipcc not found")
    out = tmp_path_factory.mktemp("asm") / "kernels.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out),
                           os.path.join(SRC, "kernels.hip")], cwd=SRC)
    return out.read_text()


def test_no_instruction_touches_an_agpr_that_is_in_flight(device_asm):
    checked = 0
    for name, lines in _functions(device_asm):
        if "dyn_phase_kernel" not in name and "rom_phase_kernel" not in name:
            continue
        bad, n_loads = _replay(lines)
        assert n_loads >= 10, (name, n_loads)     # the asm loads are there (prologue + loop)
        assert not bad, "%s:\n%s" % (name, "\n".join("  line %d: %s  <- %s" % b for b in bad[:10]))
        checked += 1
    assert checked >= 12   # 2 kernels x (NIT variants) x 3 flag variants


def test_the_model_catches_a_premature_read():
    lines = ["\tglobal_load_dwordx4 a[0:3], v[0:1], off", "\tglobal_store_dwordx4 v2, v[4:7], s[0:1]",
             "\tv_accvgpr_read_b32 v9, a2", "\ts_waitcnt 0xf71", "\tv_accvgpr_read_b32 v9, a2", "\ts_endpgm"]
    bad, n = _replay(lines)
    assert n == 1 and len(bad) == 1 and bad[0][0] == 2
    ok, _ = _replay(lines[:2] + lines[3:])
    assert not ok
