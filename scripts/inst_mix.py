"""Static instruction mix of a solve kernel per source function (VALU / SALU / LDS / memory), through the inline stacks of the line tables:
which code the vector ALU issue slots go to.  Run scripts/code_size.py first (it builds /tmp/fh_code_size/dev.elf and dis.txt).
   python scripts/code_size.py solve_kernelILi10ELb1 > /dev/null && python scripts/inst_mix.py solve_kernelILi10ELb1"""
import collections, re, subprocess, sys
sym = sys.argv[1] if len(sys.argv) > 1 else "solve_kernelILi10ELb1"
T = "/tmp/fh_code_size"
lines = open(T + "/dis.txt").read().split("\n")
start = [i for i, l in enumerate(lines) if sym in l and l.endswith(">:")][0]
end = [i for i, l in enumerate(lines[start + 1:], start + 1) if re.match(r"^[0-9a-f]+ <", l)][0]
addrs, ops = [], []
for i in range(start, end):
    m = re.match(r"^\s+(\S+).*?// ([0-9A-Fa-f]+):", lines[i])
    if m and not lines[i].startswith(";"):
        ops.append(m.group(1)); addrs.append(m.group(2))
out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-symbolizer", "-e", T + "/dev.elf", "--inlines", "--functions=short", "--no-demangle"],
                     input="\n".join("0x" + a for a in addrs) + "\n", capture_output=True, text=True).stdout
blocks = [b for b in out.split("\n\n") if b.strip()]
assert len(blocks) == len(ops)
BUCKETS = ["moments", "compute_states", "scan", "build_g", "project", "backsolve", "add_row", "drop_row", "bind_assignment", "qp_loop", "qp_run", "analyze",
           "snapshot_save", "snapshot_restore", "setup_trial", "screen_constant_rows", "dt_initial", "bad_input", "init_problem", "look_around", "donate_trials",
           "donate", "take_task", "install_frame", "publish_incumbent", "finish_part", "begin_donation", "pair_glue_one", "clock_at", "choose_r_index", "search",
           "run_problem", "solve_kernel"]
stat = collections.defaultdict(collections.Counter)
for b, op in zip(blocks, ops):
    ls = b.strip().split("\n")
    frames = [ls[i] for i in range(0, len(ls), 2)]  # innermost first
    k = next((bk for f in frames for bk in BUCKETS if f.startswith(bk)), frames[-1])
    stat[k]["valu" if op.startswith("v_") else ("salu" if op.startswith("s_") else ("lds" if op.startswith("ds_") else "mem"))] += 1
print("%-24s %6s %6s %5s %5s   (static instructions; a function inlined at two sites counts twice)" % ("function", "VALU", "SALU", "LDS", "MEM"))
tot = collections.Counter()
for k, c in sorted(stat.items(), key=lambda kv: -sum(kv[1].values())):
    print("%-24s %6d %6d %5d %5d" % (k, c["valu"], c["salu"], c["lds"], c["mem"]))
    tot.update(c)
print("total", dict(tot))
