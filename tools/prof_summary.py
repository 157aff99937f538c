"""Per-step kernel time table from one or two rocprofv3 kernel_stats CSVs (gpurun_out/<name>_kernel_stats.csv):
    python tools/prof_summary.py r02_b [r02_a] [--steps 40]"""
import csv
import re
import sys


def short(name):
    m = re.search(r"(?:\(anonymous namespace\)::|at::native::)?([A-Za-z_0-9]+)(<[^(>]*>)?\(", name)
    if not m:
        return name[:50]
    base, targ = m.group(1), m.group(2) or ""
    if base in ("vectorized_elementwise_kernel", "elementwise_kernel_manual_unroll", "unrolled_elementwise_kernel", "reduce_kernel"):
        f = re.search(r"at::native::(\w+Functor|\w+_kernel_cuda|\w+Ops)", name)
        targ = "<" + (f.group(1) if f else "") + ">"
    return (base + targ)[:52]


def load(tag, steps):
    out = {}
    rows = list(csv.DictReader(open(f"gpurun_out/{tag}_kernel_stats.csv")))
    one_per_step = [int(r["Calls"]) for r in rows if "color_forward_x3_kernel<true>" in r["Name"] or "color_forward_kernel<true>" in r["Name"]]
    if one_per_step:
        steps = one_per_step[0]          # the colour trunk runs once per optimisation step: the trace's own step count
    for r in rows:
        k = short(r["Name"])
        c, t = out.get(k, (0.0, 0.0))
        out[k] = (c + int(r["Calls"]) / steps, t + int(r["TotalDurationNs"]) / steps / 1e3)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 40
    a = load(args[0], steps)
    b = load(args[1], steps) if len(args) > 1 else None
    print(f"{'kernel':52s} {'calls':>6s} {'us/step':>9s}" + (f" {'(' + args[1] + ')':>10s}" if b else ""))
    for k, (c, t) in sorted(a.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"{k:52s} {c:6.1f} {t:9.1f}" + (f" {b.get(k, (0, 0))[1]:10.1f}" if b else ""))
    print(f"{'TOTAL':52s} {sum(c for c, _ in a.values()):6.1f} {sum(t for _, t in a.values()):9.1f}" +
          (f" {sum(t for _, t in b.values()):10.1f}" if b else ""))


if __name__ == "__main__":
    main()
