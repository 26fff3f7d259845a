"""Counts of an instruction pattern per (innermost, outermost-known) source function in a solve kernel.
   python scripts/r6/isa_count.py <sym> <regex>"""
import collections, re, subprocess, sys
sym, pat = sys.argv[1], re.compile(sys.argv[2])
T = "/tmp/fh_code_size"
lines = open(T + "/dis.txt").read().split("\n")
start = [i for i, l in enumerate(lines) if sym in l and l.endswith(">:")][0]
end = ([i for i, l in enumerate(lines[start + 1:], start + 1) if re.match(r"^[0-9a-f]+ <", l)] + [len(lines)])[0]
addrs, ops = [], []
for i in range(start, end):
    m = re.match(r"^\s+(\S.*?)\s+// ([0-9A-Fa-f]+):", lines[i])
    if m and not lines[i].startswith(";") and pat.search(m.group(1)):
        ops.append(m.group(1)); addrs.append(m.group(2))
out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-symbolizer", "-e", T + "/dev.elf", "--inlines", "--functions=short", "--no-demangle"],
                     input="\n".join("0x" + a for a in addrs) + "\n", capture_output=True, text=True).stdout
blocks = [b for b in out.split("\n\n") if b.strip()]
c = collections.Counter()
for b in blocks:
    ls = b.strip().split("\n")
    frames = [ls[i] for i in range(0, len(ls), 2)]
    line = ls[1].rsplit("/", 1)[-1]
    c[(" < ".join(f[:20] for f in frames[:4]), line)] += 1
for k, v in c.most_common(60): print(v, k)
print("total", sum(c.values()))
