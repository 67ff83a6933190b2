"""ISA-level guard for the gfx950 store-data hazard ("Write-back hazard"; runs without a GPU).

Measured on MI355X (tools/store_hazard_probe.hip, profiles/r02_store_hazard_probe.json): a VALU write to the first
data register of a `buffer_store_dwordx4` lands in the stored data of lanes 12-15 of every 16-lane row unless at
least ONE wait state separates the two when the store's soffset is an SGPR, TWO when it is an immediate.  hipcc pads
the immediate form itself and emits nothing for the SGPR form (llvm GCNHazardRecognizer::createsVALUHazard), so the
library must not contain the SGPR form at all -- that is what produced round 1's "wrong factors under the
look-ahead".  This test disassembles every gfx950 code object of the built library and checks
  (1) no MUBUF store of more than 64 bits uses an SGPR soffset;
  (2) behind every >64-bit buffer/global store, the first VALU write that overlaps its data registers is at least two
      wait states away.
"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "projected-lmc_amd", "projectedlmc", "libplmc_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

STORE = re.compile(r"^\s*(buffer_store_dwordx[34]|buffer_store_format_xyzw?|global_store_dwordx[34]|flat_store_dwordx[34])\s+(.*)$")
REGRANGE = re.compile(r"v\[(\d+):(\d+)\]")


def _disassemble():
    if not (os.path.exists(LIB) and os.path.exists(OBJDUMP)):
        pytest.skip("library or llvm-objdump not available")
    tmp = tempfile.mkdtemp(prefix="plmc_isa_")
    try:
        lib = shutil.copy(LIB, tmp)
        subprocess.run([OBJDUMP, "--offloading", lib], cwd=tmp, capture_output=True, check=True)
        text = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" in f:
                out = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], capture_output=True, text=True, check=True)
                text.append(out.stdout)
        return "\n".join(text)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


@pytest.fixture(scope="module")
def isa():
    lines = []
    func = "?"
    for raw in _disassemble().splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", raw)
        if m:
            func = m.group(1)
            continue
        ins = raw.split("//")[0].strip()
        if ins and not ins.endswith(":") and re.match(r"^[a-z]", ins):
            lines.append((func, ins))
    assert len(lines) > 1000, "disassembly came out empty"
    return lines


def _data_regs(mnemonic, operands):
    ops = [o.strip() for o in operands.split(",")]
    # buffer_store: vdata, vaddr, srsrc, soffset ...;  global_store: vaddr, vdata, saddr
    data = ops[0] if mnemonic.startswith("buffer") else ops[1]
    m = REGRANGE.match(data)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def _valu_dst(ins):
    """registers written by a VALU instruction (first operand), empty for everything else"""
    if not ins.startswith("v_") or ins.startswith("v_cmp") or ins.startswith("v_nop"):
        return set()
    dst = ins.split(None, 1)[1].split(",")[0].strip() if " " in ins else ""
    m = REGRANGE.match(dst)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", dst)
    return {int(m.group(1))} if m else set()


def test_no_wide_buffer_store_with_sgpr_soffset(isa):
    bad = []
    for func, ins in isa:
        m = STORE.match(ins)
        if m and m.group(1).startswith("buffer"):
            ops = [o.strip() for o in m.group(2).split(",")]
            soffset = ops[3].split()[0]
            if re.match(r"^(s\d+|m0|ttmp\d+)$", soffset):
                bad.append((func, ins))
    assert not bad, "MUBUF store > 64 bits with an SGPR soffset (unpadded store-data hazard on gfx950): %r" % bad[:5]


def test_wide_stores_are_padded_against_valu_overwrite(isa):
    bad, seen = [], 0
    for i, (func, ins) in enumerate(isa):
        m = STORE.match(ins)
        if not m:
            continue
        seen += 1
        regs = _data_regs(m.group(1), m.group(2))
        waits = 0
        for _, nxt in isa[i + 1:i + 4]:
            if waits >= 2:
                break
            if _valu_dst(nxt) & regs:
                bad.append((func, ins, nxt, waits))
                break
            mm = re.match(r"s_nop\s+(\d+)", nxt)
            waits += int(mm.group(1)) + 1 if mm else 1
    assert seen > 100, "no wide stores found: the scan is not looking at the tile kernels"
    assert not bad, "VALU overwrite of store data inside the hazard window: %r" % bad[:5]
