"""Static VALU / SALU / LDS / MEM counts of a solve kernel per INLINE PATH (outermost known function > ... > bucket), so that the instance
of a function inside the active-set loop is told from its cold copies.   python scripts/r6/inst_paths.py <sym>"""
import collections, re, subprocess, sys
sym = sys.argv[1]
T = "/tmp/fh_code_size"
lines = open(T + "/dis.txt").read().split("\n")
start = [i for i, l in enumerate(lines) if sym in l and l.endswith(">:")][0]
end = ([i for i, l in enumerate(lines[start + 1:], start + 1) if re.match(r"^[0-9a-f]+ <", l)] + [len(lines)])[0]
addrs, ops = [], []
for i in range(start, end):
    m = re.match(r"^\s+(\S+).*?// ([0-9A-Fa-f]+):", lines[i])
    if m and not lines[i].startswith(";"):
        ops.append(m.group(1)); addrs.append(m.group(2))
out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-symbolizer", "-e", T + "/dev.elf", "--inlines", "--functions=short", "--no-demangle"],
                     input="\n".join("0x" + a for a in addrs) + "\n", capture_output=True, text=True).stdout
blocks = [b for b in out.split("\n\n") if b.strip()]
KEEP = ["moments", "compute_states", "scan", "build_g", "project", "backsolve", "add_row", "drop_row", "bind_assignment", "qp_loop", "qp_run", "analyze",
        "snapshot_save", "snapshot_restore", "setup_trial", "screen_constant_rows", "dt_initial_exact", "dt_initial", "bad_input", "init_problem", "look_around", "donate_trials",
        "donate", "take_task", "install_frame", "publish_incumbent", "finish_part", "begin_donation", "pair_glue_one", "clock_at", "choose_r_index", "search",
        "run_problem", "solve_kernel", "row_norms", "unit_done", "flush_done", "note_limited", "x0_outside_box"]
stat = collections.defaultdict(collections.Counter)
for b, op in zip(blocks, ops):
    ls = b.strip().split("\n")
    frames = [ls[i] for i in range(0, len(ls), 2)]  # innermost first
    path = []
    for f in reversed(frames):
        k = next((bk for bk in KEEP if f.startswith(bk)), None)
        if k and (not path or path[-1] != k): path.append(k)
    key = ">".join(path[-3:]) if path else frames[-1][:30]
    stat[key]["valu" if op.startswith("v_") else ("salu" if op.startswith("s_") else ("lds" if op.startswith("ds_") else "mem"))] += 1
print("%-60s %6s %6s %5s %5s" % ("path", "VALU", "SALU", "LDS", "MEM"))
tot = collections.Counter()
for k, c in sorted(stat.items(), key=lambda kv: -kv[1]["valu"]):
    print("%-60s %6d %6d %5d %5d" % (k, c["valu"], c["salu"], c["lds"], c["mem"]))
    tot.update(c)
print("total", dict(tot))
