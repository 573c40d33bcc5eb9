"""Attribute the unsymbolised stack of profiles/r03_pmc_full_pass_sigsegv.log to a shared library by the SPACING of its frames.
Frames 5-8 of the trace (0x...55b266, ...55a7e7, ...57427b, ...5d6cf1, the last one directly under start_thread) lie within 0x80000
of each other: one library.  A library is mapped page-aligned, so a return address keeps its low 12 bits and the distances
between return addresses are those of the file.  For every candidate library: collect the addresses that FOLLOW a call
instruction (objdump -d), and look for a quadruple with exactly these low bits and distances.
    python scripts/match_frames.py [lib ...]"""
import re, subprocess, sys
FRAMES = [0x55b266, 0x55a7e7, 0x57427b, 0x5d6cf1]          # relative to an unknown, page-aligned base
LIBS = sys.argv[1:] or ["/opt/rocm/lib/librocprofiler-sdk.so.1.1.0", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so.1.1.0",
                        "/opt/rocm/lib/libhsa-runtime64.so.1.18.70200", "/opt/rocm/lib/libamdhip64.so.7.2.70200"]
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
for lib in LIBS:
    out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", lib], capture_output=True, text=True).stdout
    rets, prev_call, func, funcs = set(), False, None, {}
    cur = None
    for line in out.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(.*)>:$", line)
        if m:
            cur = m.group(2); prev_call = False; continue
        m = re.match(r"^\s*([0-9a-f]+):\s+(\S+)", line)
        if not m: continue
        addr = int(m.group(1), 16)
        if prev_call:
            rets.add(addr); funcs[addr] = cur
        prev_call = m.group(2).startswith("call")
    hits = []
    for r in rets:
        if (r & 0xfff) != (FRAMES[0] & 0xfff): continue
        if all((r + f - FRAMES[0]) in rets for f in FRAMES[1:]):
            hits.append(r)
    print("%s: %d return addresses, %d matching quadruples" % (lib, len(rets), len(hits)))
    for r in hits:
        for f in FRAMES:
            a = r + f - FRAMES[0]
            print("    frame +0x%x -> file address 0x%x in %s" % (f, a, funcs.get(a)))
