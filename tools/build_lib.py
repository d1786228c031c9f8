#!/usr/bin/env python3
"""Build libnbp.so for gfx950: the kernel groups (csrc/nbp_k_*.hip) and the host side (nbp_api.hip, nbp_host.cpp)
are compiled in parallel and linked into one shared library.

    tools/build_lib.py                      # incremental build of csrc/libnbp.so
    tools/build_lib.py --resources FILE     # also write the per-kernel VGPR / scratch / occupancy table
                                            # (-Rpass-analysis=kernel-resource-usage) to FILE
    tools/build_lib.py --single -DNBP_PHASE_TIMING -o tools/libnbp_dbg.so   # one translation unit (debug builds whose
                                            # device-side globals must be shared by all kernels)
"""
import argparse
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "incrementalinference.jl_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", os.environ.get("NBP_FP_CONTRACT", "-ffp-contract=off")]
# -ffp-contract=off: ONE ROUNDING PER WRITTEN OPERATION in every kernel -- the arithmetic of the CPU checker (gcc -ffp-contract=off)
# and of Julia; the multiply-adds that are wanted are explicit fma() calls (round 6: DESIGN.md section 5; +1..2 % per solve,
# profiles/r06_contract_off_cost.txt)
HEADERS = [os.path.join(CSRC, h) for h in ("nbp_kernels.h", "nbp_device.h", "nbp_lcv_table.h", "nbp_fused.h")] + \
          [os.path.join(ROOT, "include", h) for h in ("nbp.h", "nbp_host.h", "nbp_math.h")]


def sources():
    ks = sorted(f for f in os.listdir(CSRC) if f.startswith("nbp_k_") and f.endswith(".hip"))
    return [os.path.join(CSRC, f) for f in ks] + [os.path.join(CSRC, "nbp_api.hip"), os.path.join(CSRC, "nbp_host.cpp")]


def parse_resources(text):
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    if not rows:
        return []
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        r["demangled"] = re.sub(r"^void ", "", n).split("(")[0]
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-o", "--out", default=os.path.join(CSRC, "libnbp.so"))
    ap.add_argument("--resources", default=None)
    ap.add_argument("--single", action="store_true")
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--objdir", default=os.path.join(CSRC, "build"))
    args, extra = ap.parse_known_args()
    t0 = time.time()
    if args.single:
        cmd = [HIPCC] + FLAGS + ["-shared", "-DNBP_TU=0xFFFF"] + extra + [os.path.join(CSRC, "nbp_api.hip"), os.path.join(CSRC, "nbp_host.cpp"), "-o", args.out]
        subprocess.check_call(cmd)
        print(f"built {args.out} (single translation unit) in {time.time() - t0:.0f} s")
        return
    os.makedirs(args.objdir, exist_ok=True)
    tag = ("_" + re.sub(r"\W", "", "".join(extra))) if extra else ""
    # what a translation unit includes: the kernel groups everything but the host header, nbp_host.cpp only the two ABI headers
    def hdr_m(src):
        hs = [h for h in HEADERS if os.path.exists(h)]
        if src.endswith("nbp_host.cpp"):
            hs = [h for h in hs if os.path.basename(h) in ("nbp.h", "nbp_host.h")]  # (no device code: nbp_math.h is not included)
        elif os.path.basename(src).startswith("nbp_k_"):
            hs = [h for h in hs if os.path.basename(h) != "nbp_host.h"]
        return max(os.path.getmtime(h) for h in hs)
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(args.objdir, os.path.basename(src).rsplit(".", 1)[0] + tag + ".o")
        log = obj + ".log"
        objs.append((src, obj, log))
        dep_m = max(os.path.getmtime(src), hdr_m(src))
        want_res = args.resources and src.endswith(".hip") and "nbp_k_" in src
        if not args.force and os.path.exists(obj) and os.path.getmtime(obj) >= dep_m and (not want_res or os.path.exists(log)):
            continue
        own = []  # per-file flags: a `// hipcc-flags: ...` line in the source
        for line in open(src):
            if line.startswith("// hipcc-flags:"):
                own += line.split(":", 1)[1].split()
        cmd = [HIPCC] + FLAGS + own + extra + ["-c", src, "-o", obj]
        if src.endswith(".hip"):
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        jobs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True), log))
    failed = False
    for src, p, log in jobs:
        _, err = p.communicate()
        open(log, "w").write(err)
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src} failed ---\n" + "\n".join(l for l in err.splitlines() if "remark:" not in l and not l.startswith("  ") and "| ^" not in l)[-6000:] + "\n")
    if failed:
        sys.exit(1)
    if jobs or not os.path.exists(args.out) or any(os.path.getmtime(o) > os.path.getmtime(args.out) for _, o, _ in objs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for _, o, _ in objs] + ["-o", args.out])
    print(f"built {args.out}: {len(jobs)} of {len(objs)} translation units compiled in {time.time() - t0:.0f} s")
    if args.resources:
        rows = []
        for src, obj, log in objs:
            if os.path.exists(log):
                for r in parse_resources(open(log).read()):
                    r["tu"] = os.path.basename(src)
                    rows.append(r)
        rows.sort(key=lambda r: r["demangled"])
        with open(args.resources, "w") as f:
            f.write("# per-kernel resources of libnbp.so, gfx950 (hipcc -O3 -Rpass-analysis=kernel-resource-usage; written by tools/build_lib.py --resources)\n")
            f.write(f"# {'kernel':58s} {'VGPRs':>5s} {'AGPRs':>5s} {'SGPRs':>5s} {'scratch B/lane':>14s} {'waves/SIMD':>10s} {'SGPR spills':>11s} {'VGPR spills':>11s}  translation unit\n")
            for r in rows:
                f.write(f"{r['demangled']:60s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} {r.get('ScratchSize [bytes/lane]', '?'):>14s} "
                        f"{r.get('Occupancy [waves/SIMD]', '?'):>10s} {r.get('SGPRs Spill', '?'):>11s} {r.get('VGPRs Spill', '?'):>11s}  {r['tu']}\n")
        print(f"wrote {args.resources} ({len(rows)} kernels)")


if __name__ == "__main__":
    main()
