"""Where the bytes of a solve kernel are: code size per source function (device ISA attributed through the line tables).
   python scripts/code_size.py 'solve_kernelILi10ELb1' [more symbol substrings...]
Builds fh_capi.hip for gfx950 with -gline-tables-only into /tmp, disassembles it with llvm-objdump -d -l and sums the instruction
bytes by the source function whose line range holds the instruction's (innermost inlined) line."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TMP = "/tmp/fh_code_size"
os.makedirs(TMP, exist_ok=True)
LLVM = "/opt/rocm/lib/llvm/bin"
if "--reuse" not in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-sink-insts-to-avoid-spills", "-mllvm",
                           "-disable-machine-licm", "-gline-tables-only", "-c", "--cuda-device-only", "-o", TMP + "/fh_capi.co"] +
                          os.environ.get("FASTERHIP_EXTRA_FLAGS", "").split() + [ROOT + "/faster_amd/csrc/fh_capi.hip"], stderr=subprocess.DEVNULL)
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + TMP + "/fh_capi.co",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + TMP + "/dev.elf"])
    with open(TMP + "/dis.txt", "w") as f:
        subprocess.check_call([LLVM + "/llvm-objdump", "-d", "-l", "--no-show-raw-insn", TMP + "/dev.elf"], stdout=f)

# function line ranges of the sources (a crude scan: "__device__ ... name(" at brace depth <= 2 starts a function)
def functions(path):
    out, name, start = [], None, 0
    for i, l in enumerate(open(path), 1):
        m = re.match(r"\s*(?:template\s*<[^>]*>\s*)?(?:static\s+)?(?:__device__|__global__|__host__)[^;{(]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", l)
        if m and not l.strip().startswith("//"):
            if name: out.append((start, i - 1, name))
            name, start = m.group(1), i
    if name: out.append((start, 10 ** 9, name))
    return out
srcs = {}
def owner(path, line):
    b = os.path.basename(path)
    if b not in srcs:
        srcs[b] = functions(path) if os.path.exists(path) and "/repo/" in path else None
    if srcs[b] is None: return "<" + b + ">"
    for s, e, n in srcs[b]:
        if s <= line <= e: return b.split(".")[0] + ":" + n
    return "<" + b + ">"

want = [a for a in sys.argv[1:] if not a.startswith("--")] or ["solve_kernelILi10ELb1"]
sym, cur, prev_addr, prev_owner = None, None, None, None
size = collections.defaultdict(lambda: collections.Counter())
for l in open(TMP + "/dis.txt"):
    m = re.match(r"^([0-9a-f]+) <(.*)>:", l)
    if m:
        sym, prev_addr = m.group(2), None
        continue
    m = re.match(r"^; (/\S+):(\d+)", l)
    if m:
        cur = owner(m.group(1), int(m.group(2)))
        continue
    m = re.match(r"^\s+(\S.*?)\s+// ([0-9A-Fa-f]+):", l)
    if m and sym:
        addr = int(m.group(2), 16)
        if prev_addr is not None: size[sym][prev_owner] += addr - prev_addr
        prev_addr, prev_owner = addr, cur
for s, c in size.items():
    if not any(w in s for w in want): continue
    tot = sum(c.values())
    print("%s: %d bytes" % (s[:90], tot))
    for k, v in c.most_common(40): print("   %-44s %7d  %5.1f%%" % (k, v, 100.0 * v / tot))
