"""Names the frames of the round-2 profiled-stress crashes (gpurun_out/stress_kt_{1,3}.log) without the process' memory map.

    python scratch/resolve_stress_frames.py <librlr_gpu.so built at commit 598c197> [log ...]

The logs hold raw return addresses; only the two frames inside librlr_gpu.so carry names.  Load bases are unknown (ASLR), but
bases are page aligned and the distance between two frames of ONE library is fixed, so a pair of frames is identified by
    (low 12 bits of frame A, low 12 bits of frame B, B - A)
matched against the disassembly of a candidate library: A and B must both be return addresses (the instruction before is a
call) -- or, for the faulting PC, an instruction boundary.  Candidates: /opt/rocm/lib/librocprofiler-sdk.so (the rocprofv3
tool's SDK) and /opt/rocm/lib/libhsa-runtime64.so.1 (ROCr, which the preloaded tool pulls in ahead of any other copy).
Findings are written up in profiles/r03_profiled_stress_aborts.md.
"""
import re
import subprocess
import sys

SDK = "/opt/rocm/lib/librocprofiler-sdk.so.1.1.0"
HSA = "/opt/rocm/lib/libhsa-runtime64.so.1"
INSN = re.compile(r"^\s*([0-9a-f]+):\t(.*)$")


def disasm(path):
    out = subprocess.run(["objdump", "-d", "-C", "--no-show-raw-insn", path], capture_output=True, text=True, check=True).stdout
    insns = []
    for line in out.splitlines():
        m = INSN.match(line)
        if m:
            insns.append((int(m.group(1), 16), m.group(2)))
    return insns, {a: i for i, (a, _) in enumerate(insns)}


def is_ret_addr(insns, amap, a):
    i = amap.get(a)
    return i is not None and i > 0 and insns[i - 1][1].startswith("call")


def match_pair(insns, amap, lo_a, lo_b, delta, a_must_be_ret):
    """all (A, B) with A & 0xfff == lo_a, B & 0xfff == lo_b, B - A == delta, B a return address."""
    hits = []
    for b, _ in insns:
        if (b & 0xFFF) != lo_b or not is_ret_addr(insns, amap, b):
            continue
        a = b - delta
        if a in amap and (a & 0xFFF) == lo_a and (not a_must_be_ret or is_ret_addr(insns, amap, a)):
            hits.append((a, b))
    return hits


def frames_of(log):
    pc, frames, named = None, [], {}
    for line in open(log, errors="replace"):
        m = re.match(r"PC: @\s+0x([0-9a-f]+)", line)
        if m:
            pc = int(m.group(1), 16)
        m = re.match(r"\s+@\s+0x([0-9a-f]+)\s+(.*)$", line)
        if m:
            frames.append(int(m.group(1), 16))
            if m.group(2).startswith("rlr"):
                named[int(m.group(1), 16)] = m.group(2)
    return pc, frames, named


def main():
    so = sys.argv[1]
    logs = sys.argv[2:] or ["gpurun_out/stress_kt_1.log", "gpurun_out/stress_kt_3.log"]
    lib_insns, lib_map = disasm(so)
    sdk_insns, sdk_map = disasm(SDK)
    hsa_insns, hsa_map = disasm(HSA)
    for log in logs:
        pc, frames, named = frames_of(log)
        print(f"== {log}: PC {pc:#x}")
        # 1. the HIP call: the named librlr frames are return addresses right after `call hipXxx@plt`
        in_lib = [f for f in frames if f in named]  # innermost first
        inner, outer = in_lib[0], in_lib[1]
        for a, _ in lib_insns:
            if (a & 0xFFF) == (inner & 0xFFF) and is_ret_addr(lib_insns, lib_map, a):
                b = a + (outer - inner)
                if b in lib_map and is_ret_addr(lib_insns, lib_map, b):
                    print(f"   {named[inner].split('(')[0]}: return address {a:#x} after `{lib_insns[lib_map[a] - 1][1]}`")
                    print(f"   {named[outer].split('(')[0]}: return address {b:#x} after `{lib_insns[lib_map[b] - 1][1][:90]}`")
        # 2. the frame the library calls into (the HIP entry as the process sees it) and the faulting PC: one library?
        entry = frames[frames.index(inner) - 1]
        for a, b in match_pair(sdk_insns, sdk_map, pc & 0xFFF, entry & 0xFFF, entry - pc, a_must_be_ret=False):
            if sdk_insns[sdk_map[b] - 1][1].startswith("call   *"):
                print(f"   librocprofiler-sdk: faulting instruction {a:#x} `{sdk_insns[sdk_map[a]][1]}`;"
                      f" HIP-API wrapper frame returns to {b:#x} after `{sdk_insns[sdk_map[b] - 1][1][:40]}`")
        # 3. the two frames between libamdhip64's and the PC
        f1, f2 = frames[frames.index(pc) + 1], frames[frames.index(pc) + 2]
        for a, b in match_pair(hsa_insns, hsa_map, f2 & 0xFFF, f1 & 0xFFF, f1 - f2, a_must_be_ret=True):
            print(f"   libhsa-runtime64 (ROCr 7.2.0): {a:#x} after `{hsa_insns[hsa_map[a] - 1][1]}`  ->  {b:#x} after"
                  f" `{hsa_insns[hsa_map[b] - 1][1]}` (the interceptor callback)")


if __name__ == "__main__":
    main()
