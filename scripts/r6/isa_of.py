"""ISA of one source function as it was inlined into a solve kernel (through the inline stacks of the line tables).
   python scripts/code_size.py <sym> > /dev/null && python scripts/r6/isa_of.py <sym> <function> [first-instance-only=1]
Prints the instructions whose inline stack contains <function>, with the innermost frame's name and source line."""
import re, subprocess, sys
sym, fn = sys.argv[1], sys.argv[2]
T = "/tmp/fh_code_size"
lines = open(T + "/dis.txt").read().split("\n")
start = [i for i, l in enumerate(lines) if sym in l and l.endswith(">:")][0]
end = ([i for i, l in enumerate(lines[start + 1:], start + 1) if re.match(r"^[0-9a-f]+ <", l)] + [len(lines)])[0]
addrs, ops = [], []
for i in range(start, end):
    m = re.match(r"^\s+(\S.*?)\s+// ([0-9A-Fa-f]+):", lines[i])
    if m and not lines[i].startswith(";"):
        ops.append(m.group(1)); addrs.append(m.group(2))
out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-symbolizer", "-e", T + "/dev.elf", "--inlines", "--functions=short", "--no-demangle"],
                     input="\n".join("0x" + a for a in addrs) + "\n", capture_output=True, text=True).stdout
blocks = [b for b in out.split("\n\n") if b.strip()]
assert len(blocks) == len(ops)
prev = False
n = 0
for a, op, b in zip(addrs, ops, blocks):
    ls = b.strip().split("\n")
    frames = [(ls[i], ls[i + 1].rsplit("/", 1)[-1]) for i in range(0, len(ls), 2)]
    hit = any(f[0].startswith(fn) for f in frames)
    if hit:
        if not prev: print("---- @%s" % a)
        print("%-70s ; %s %s" % (op[:70], frames[0][0][:24], frames[0][1]))
        n += 1
    prev = hit
print("total", n)
