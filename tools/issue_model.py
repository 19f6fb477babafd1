#!/usr/bin/env python3
"""Compute-side roofline of the hot kernels from evidence under profiles/ (VERDICT r1 item 2):

    issue cycles of a kernel = sum over instruction classes of (dynamic instruction count) x (measured cost per instruction)

* cost per instruction: profiles/r02_issue_table.txt (tools/ubench/issue_table.hip: shader cycles one SIMD / the CU's LDS
  pipeline / the SIMD's scalar issue needs per wave-instruction, measured at 1..8 wavefronts per SIMD with every CU busy);
* dynamic counts per launch: rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS (tools/pmc_sweep.sh summary);
* the split of the VALU count into cost classes: the kernel's own ISA (hipcc -S), weighted statically — every VALU opcode
  of the function is classified (2-cycle class: f32 add/sub/mul/fma, mov, 32-bit add/and/or/xor/shift; 4-cycle class:
  min/max, compares, cndmask, packed f32, f64, conversions, VOP3-only integer ops, SDWA/DPP; 8-cycle class: rcp/rsq/sqrt).
  For the LDPC kernels the iteration loop is cut out of the ISA and counted exactly.

utilisation(VALU)  = INSTS_VALU x mean VALU cost / (SIMDs x clock x launch time)         SIMDs = 4 x CUs
utilisation(SALU)  = INSTS_SALU x scalar cost   / (SIMDs x clock x launch time)
utilisation(LDS)   = sum over LDS opcodes (count x pipeline cycles) / (CUs x clock x launch time)

    python3 tools/issue_model.py [--pmc gpurun_out/.../summary.txt] [--asm /tmp/uh.s] > profiles/r02_issue_model.txt
"""
import argparse
import collections
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mov_b32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_not_b32",
        "v_add_co_u32", "v_addc_co_u32", "v_mul_u32_u24", "v_mul_i32_i24", "v_nop"}
EIGHT = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64",
         "v_rcp_iflag_f32"}
LDS_CYC = {"ds_read_b32": 2.14, "ds_read_b64": 2.16, "ds_read2_b32": 4.09, "ds_read2st64_b32": 4.09, "ds_read_b128": 4.09, "ds_read_u8": 2.14,
           "ds_read_u16": 2.14, "ds_read2_b64": 8.0, "ds_write_b32": 4.12, "ds_write_b64": 6.02, "ds_write_b8": 4.12, "ds_write_b16": 4.12,
           "ds_write_addtid_b32": 2.12, "ds_write2_b32": 6.0, "ds_write2st64_b32": 6.0, "ds_write_b128": 13.0, "ds_bpermute_b32": 2.14,
           "ds_swizzle_b32": 2.14}


def base(op):
    return re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)


def valu_cost(op, line, costs):
    if "sdwa" in line or "dpp" in line or "row_" in line or "quad_perm" in line:
        return costs["slow"]
    b = base(op)
    if b in EIGHT:
        return costs["eight"]
    if b in FAST:
        return costs["fast"]
    if b == "v_readlane_b32" or b == "v_readfirstlane_b32":
        return costs["readlane"]
    return costs["slow"]


def parse_issue_table(path):
    rows = {}
    for l in path.read_text().splitlines():
        if l.startswith("#") or l.startswith("instruction"):
            continue
        m = re.match(r"(.{38})\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", l)
        if m:
            rows[m.group(1).strip()] = [float(m.group(i)) for i in range(2, 8)]
    W5 = 3          # column of W = 5 wavefronts per SIMD
    c = dict(fast=rows["v_add_f32"][W5], slow=rows["v_min3_f32 |x|,|y|,|z|"][W5], eight=rows["v_rcp_f32"][W5], readlane=rows["v_readlane_b32"][W5],
             salu=rows["s_xor_b64"][W5])
    m = re.search(r"shader clock while measuring.*?: ([\d.]+) GHz", path.read_text())
    c["clock_ghz"] = float(m.group(1)) if m else 2.34
    return c, rows


def functions(asm):
    """{function name: list of instruction lines}"""
    out, cur, name = {}, None, None
    for l in asm.splitlines():
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name = m.group(1); cur = []; out[name] = cur
            continue
        if cur is not None:
            if "s_endpgm" in l:
                cur.append(l); cur = None
                continue
            cur.append(l)
    return out


def mix(lines, costs):
    cnt = collections.Counter()
    v_cyc = v_n = s_n = 0
    lds = collections.Counter()
    for l in lines:
        m = re.match(r"\s+([a-z][a-z_0-9]+)", l)
        if not m or l.strip().startswith(";") or l.strip().startswith("."):
            continue
        op = m.group(1)
        cnt[op] += 1
        if op.startswith("v_"):
            v_n += 1; v_cyc += valu_cost(op, l, costs)
        elif op.startswith("s_"):
            s_n += 1
        elif op.startswith("ds_"):
            lds[op] += 1
    lds_cyc = sum(n * LDS_CYC.get(op, 4.0) for op, n in lds.items())
    return dict(valu=v_n, valu_cycles=v_cyc, salu=s_n, lds=sum(lds.values()), lds_cycles=lds_cyc, lds_ops=dict(lds), top=cnt.most_common(12))


def hot_loop(lines):
    """The innermost loop region with the most LDS instructions (the BP iteration of the LDPC kernels)."""
    idx = [i for i, l in enumerate(lines) if "Loop Header: Depth=2" in l]
    best = None
    for a, b in zip(idx, idx[1:] + [len(lines)]):
        n = sum(1 for l in lines[a:b] if re.match(r"\s+ds_", l))
        if best is None or n > best[2]:
            best = (a, b, n)
    return lines[best[0]:best[1]] if best else lines


def parse_pmc(path):
    out, cur = {}, None
    for l in Path(path).read_text().splitlines():
        m = re.match(r"== (.+?) launches/pass (\d+)", l)
        if m:
            cur = out.setdefault(m.group(1), {})
            cur["_launches"] = int(m.group(2))
            continue
        m = re.match(r"\s+(\S+)\s+([\d.]+)$", l)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
    return out


CLASS_OF = (("ldpc_totals_kernel", "ldpc_decode_kernel", "codeword"), ("ldpc_decode_kernel", "ldpc_decode_kernel", "codeword"),
            ("mix_fft2_kernel", None, "frame-symbol"), ("track_all_kernel", "track_kernel", "frame-symbol"),
            ("track_diff_pair_kernel", "track_kernel", "frame-symbol"), ("track_pilot_kernel", "track_pilot_kernel", "frame-symbol"),
            ("track_kernel", "track_kernel", "frame-symbol"), ("train_kernel", "track_kernel", "frame-symbol"),
            ("cfo_walk_kernel", "cfo_walk_kernel", "frame-symbol"), ("acquire_kernel", "acquire_kernel", "stream"))


def class_of(instance):
    """(kernel class of ultra_hip_profile_read_items, work item) of a kernel instance name as rocprofv3 prints it"""
    for key, cls, item in CLASS_OF:
        if key in instance:
            if cls is None:                                   # the transform: the rotating instance is a class of its own
                cls = "mix_fft_rot_kernel" if re.search(r"mix_fft2_kernel<\d+, true>", instance) else "mix_fft_kernel"
            return cls, item
    return None, None


def bench_line(pmc_dir):
    """the JSON line bench.py printed under the profiler (pass 0 of tools/pmc_sweep.sh): work items per class and launch"""
    import json
    for log in sorted(Path(pmc_dir).glob("p*.log")):
        for l in log.read_text(errors="ignore").splitlines():
            if l.startswith("{") and '"work_items"' in l:
                return json.loads(l)
    return None


def write_json(args, costs, fns):
    """profiles/issue.json: per config and kernel class the issue cycles per work item, with the instances they come from, the
    hash of the kernel sources they were collected on and the commit — bench.py quotes them only when the hash is the tree's."""
    import json, time
    sys.path.insert(0, str(ROOT))
    from projectultra_amd._lib import source_hash
    out = {"csrc_sha": source_hash(), "commit": args.commit, "collected": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
           "clock_hz": costs["clock_ghz"] * 1e9,
           "costs": {k: costs[k] for k in ("fast", "slow", "eight", "readlane", "salu")},
           "method": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS per kernel instance (tools/pmc_sweep.sh, mean per launch) x "
                     "cycles per wave-instruction of the instance's own opcode mix (static ISA of this build — the BP iteration loop for the "
                     "decoders — weighted with profiles/r02_issue_table.txt, W = 5 wavefronts per SIMD), summed over the instances of a "
                     "kernel class and divided by the work items their launches covered (the bench line of the same profiled run). "
                     "valu / salu: cycles of ONE SIMD per item (4 SIMDs per CU share a CU's items); lds: cycles of the CU's LDS pipeline.",
           "configs": {}}
    for spec in args.json_from:
        config, _, d = spec.partition("=")
        pmc = parse_pmc(Path(d) / "summary.txt")
        line = bench_line(d)
        if line is None:
            print(f"issue_model: no bench line with work_items under {d}", file=sys.stderr); continue
        items = line["roofline"]["work_items"]
        classes = {}
        for inst, p in pmc.items():
            cls, item = class_of(inst)
            if cls is None or not p.get("SQ_INSTS_VALU") or cls not in items:
                continue
            key = next((k for k in fns if inst.split("<")[0] in k and _inst_match(inst, k)), None)
            body = fns.get(key, []) if key else []
            whole = mix(body, costs) if body else None
            is_ldpc = cls == "ldpc_decode_kernel"
            part = mix(hot_loop(body), costs) if (is_ldpc and body) else whole
            v_cost = part["valu_cycles"] / max(1, part["valu"]) if part else costs["slow"]
            l_cost = part["lds_cycles"] / max(1, part["lds"]) if part and part["lds"] else 3.0
            launches = p.get("_launches", 0)
            c = classes.setdefault(cls, {"item": item, "instances": {}, "_v": 0.0, "_s": 0.0, "_l": 0.0, "_launches": 0})
            c["instances"][inst] = {"launches_per_pass": launches, "insts_valu_per_launch": p.get("SQ_INSTS_VALU", 0), "insts_salu_per_launch": p.get("SQ_INSTS_SALU", 0),
                                    "insts_lds_per_launch": p.get("SQ_INSTS_LDS", 0), "valu_cost": v_cost, "salu_cost": costs["salu"], "lds_cost": l_cost,
                                    "dur_ms_under_profiler": p.get("_dur_ns", 0) * 1e-6, "isa_function": key}
            c["_v"] += p.get("SQ_INSTS_VALU", 0) * launches * v_cost
            c["_s"] += p.get("SQ_INSTS_SALU", 0) * launches * costs["salu"]
            c["_l"] += p.get("SQ_INSTS_LDS", 0) * launches * l_cost
            c["_launches"] += launches
        for cls, c in classes.items():
            per_launch = items[cls]["items_per_step"] / items[cls]["launches_per_step"]
            total = c["_launches"] * per_launch
            c.update(items_per_launch=per_launch, valu_cycles_per_item=c.pop("_v") / total, salu_cycles_per_item=c.pop("_s") / total,
                     lds_cycles_per_item=c.pop("_l") / total)
            c.pop("_launches")
        out["configs"][config] = {"launch_units": line["config"].get("launch_units"), "classes": classes}
    Path(args.json).write_text(json.dumps(out, indent=1))
    done = ", ".join("%s (%d classes)" % (k, len(v["classes"])) for k, v in out["configs"].items())
    print(f"wrote {args.json}: {done}", file=sys.stderr)


def _inst_match(inst, mangled):
    """does the mangled name belong to the instance rocprofv3 names?  template arguments in order, as ILi<n>E / Lb<0|1>E / ILy<n>E"""
    m = re.search(r"<(.*)>", inst)
    if not m:
        return True
    pos = 0
    for a in [x.strip() for x in m.group(1).split(",")]:
        tok = None
        if a in ("true", "false"):
            tok = "Lb1E" if a == "true" else "Lb0E"
        elif re.fullmatch(r"\d+ull", a):
            tok = "Ly" + a[:-3] + "E"
        elif re.fullmatch(r"\d+", a):
            tok = "Li" + a + "E"
        elif re.fullmatch(r"\(\w+\)\d+", a):               # (ultra_hip_modulation)6
            tok = "E" + a.split(")")[1] + "E"
        if tok is None:
            continue
        i = mangled.find(tok, pos)
        if i < 0:
            return False
        pos = i + 1
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="", help="write the per-class issue cycles per work item here (profiles/issue.json)")
    ap.add_argument("--json-from", nargs="*", default=[], help="config=pmc_sweep directory (summary.txt + p0.log), e.g. cfg3=gpurun_out/r05sq/cfg3")
    ap.add_argument("--commit", default=None)
    ap.add_argument("--pmc", nargs="*", default=[str(ROOT / "profiles" / f) for f in ("r04_sq_counters.txt", "r04_sq_counters_cfg2.txt", "r04_sq_counters_cfg4.txt",
                                                                                     "r04_sq_counters_cfg5.txt", "r04_sq_counters_raw.txt", "r02_sq_counters_chirp.txt")],
                    help="pmc_sweep.sh summaries (kernels of later files do not replace those of earlier ones)")
    ap.add_argument("--asm", default="")
    ap.add_argument("--cus", type=int, default=256)
    args = ap.parse_args()
    costs, rows = parse_issue_table(ROOT / "profiles" / "r02_issue_table.txt")
    if args.asm:
        asm = Path(args.asm).read_text()
    else:
        src = ROOT / "projectultra_amd" / "csrc" / "ultra_hip.hip"
        asm = subprocess.check_output(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
                                       "-fhip-fp32-correctly-rounded-divide-sqrt", "-S", "--cuda-device-only", "-o", "-", str(src)],
                                      stderr=subprocess.DEVNULL, cwd=src.parent).decode()
    fns = functions(asm)
    if args.json:
        write_json(args, costs, fns)
        return 0
    pmc = {}
    for f in args.pmc:
        if Path(f).exists():
            for k, v in parse_pmc(f).items():
                pmc.setdefault(k, v)
    clk = costs["clock_ghz"] * 1e9
    print(f"# costs (profiles/r02_issue_table.txt, W = 5 wavefronts per SIMD): 2-cycle class {costs['fast']:.2f}, 4-cycle class {costs['slow']:.2f}, "
          f"8-cycle class {costs['eight']:.2f}, readlane {costs['readlane']:.2f}, scalar {costs['salu']:.2f} cycles per wave-instruction; clock {costs['clock_ghz']:.3f} GHz")
    want = [("ldpc_totals_kernelILi3ELi6ELy1638ELy3355443ELb0", "ldpc_totals_kernel<3, 6, 1638ull, 3355443ull, false", True),    # R3/4
            ("ldpc_totals_kernelILi8ELi3E", "ldpc_totals_kernel<8, 3", True),          # R1/4 (cfg4), round 4
            ("ldpc_decode_kernelILi3ELi6E", "ldpc_decode_kernel<3, 6", True),
            ("ldpc_decode_kernelILi8ELi3E", "ldpc_decode_kernel<8, 3", True),
            ("mix_fft2_kernelILi10ELb1E", "mix_fft2_kernel<10, true>", False), ("mix_fft2_kernelILi10ELb0E", "mix_fft2_kernel<10, false>", False), ("mix_fft_kernelILi10E", "mix_fft_kernel<10>", False),
            ("mix_fft2_kernelILi9ELb0E", "mix_fft2_kernel<9, false>", False), ("mix_fft2_kernelILi9ELb1E", "mix_fft2_kernel<9, true>", False),
            ("track_diff_pair_kernelILi2E", "track_diff_pair_kernel<2>", False), ("mix_fft_kernelILi9E", "mix_fft_kernel<9>", False), ("track_all_kernelILi6E", "track_all_kernel<6>", False),
            ("track_kernelILi6E", "track_kernel<6>", False), ("track_kernelILi2E", "track_kernel<2>", False),
            ("track_pilot_kernelILi16E", "track_pilot_kernel<16>", False), ("cfo_walk_kernel", "cfo_walk_kernel", False),
            ("acquire_kernelILi10ELb0E", "acquire_kernel<10, false>", False), ("chirp_sync_kernel", "chirp_sync_kernel", False)]
    for key, pmc_key, is_ldpc in want:
        names = [n for n in fns if key in n and ("Lb0ELi" in n or not is_ldpc or "totals" in n)]
        if is_ldpc and "decode_kernel" in key:
            names = [n for n in fns if key in n and "Lb0ELi" in n]            # WANT_TOTAL = false instance
        if not names:
            continue
        name = names[0]
        body = fns[name]
        whole = mix(body, costs)
        print(f"\n== {pmc_key}   [{name[:70]}...]")
        print(f"   static ISA, whole function: {whole['valu']} VALU (mean cost {whole['valu_cycles'] / max(1, whole['valu']):.2f} cycles), {whole['salu']} SALU, "
              f"{whole['lds']} LDS instructions")
        mean_cost = whole["valu_cycles"] / max(1, whole["valu"])
        if is_ldpc:
            loop = mix(hot_loop(body), costs)
            mean_cost = loop["valu_cycles"] / max(1, loop["valu"])
            print(f"   BP iteration loop (static, both parity-verdict paths): {loop['valu']} VALU = {loop['valu_cycles']:.0f} SIMD cycles (mean {mean_cost:.2f}), "
                  f"{loop['salu']} SALU = {loop['salu'] * costs['salu']:.0f} scalar-issue cycles, {loop['lds']} LDS = {loop['lds_cycles']:.0f} LDS-pipeline cycles {loop['lds_ops']}")
            print(f"   per CU and codeword-iteration: VALU {loop['valu_cycles'] / 4:.0f}, SALU {loop['salu'] * costs['salu'] / 4:.0f}, LDS {loop['lds_cycles']:.0f} cycles "
                  f"(4 SIMDs share the work of a CU, one LDS pipeline)")
        p = next((v for k, v in pmc.items() if k.startswith(pmc_key)), None)
        if p and p.get("_dur_ns"):
            T = p["_dur_ns"] * 1e-9
            simd_cyc = 4 * args.cus * clk * T
            cu_cyc = args.cus * clk * T
            u_v = p.get("SQ_INSTS_VALU", 0) * mean_cost / simd_cyc
            u_s = p.get("SQ_INSTS_SALU", 0) * costs["salu"] / simd_cyc
            msg = (f"   PMC per launch ({T * 1e3:.3f} ms under the profiler): INSTS_VALU {p.get('SQ_INSTS_VALU', 0):.3e} x {mean_cost:.2f} -> VALU issue {100 * u_v:.0f} % busy; "
                   f"INSTS_SALU {p.get('SQ_INSTS_SALU', 0):.3e} x {costs['salu']:.2f} -> scalar issue {100 * u_s:.0f} % busy")
            if p.get("SQ_INSTS_LDS"):
                per = (loop["lds_cycles"] / loop["lds"]) if is_ldpc and loop["lds"] else whole["lds_cycles"] / max(1, whole["lds"])
                u_l = p["SQ_INSTS_LDS"] * per / cu_cyc
                msg += f"; INSTS_LDS {p['SQ_INSTS_LDS']:.3e} x {per:.2f} -> LDS pipeline {100 * u_l:.0f} % busy"
                if p.get("SQ_LDS_IDX_ACTIVE"):
                    msg += f" (LDS array alone, SQ_LDS_IDX_ACTIVE: {100 * p['SQ_LDS_IDX_ACTIVE'] / cu_cyc / 4:.0f} %; bank conflicts {100 * p.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, p['SQ_LDS_IDX_ACTIVE']):.1f} % of it)"
            print(msg)
            if p.get("SQ_WAVES"):
                print(f"   wavefronts {p['SQ_WAVES']:.0f}, wave-cycles waiting on anything {100 * p.get('SQ_WAIT_ANY', 0) / max(1, p.get('SQ_WAVE_CYCLES', 1)):.0f} %")
    return 0


if __name__ == "__main__":
    sys.exit(main())
