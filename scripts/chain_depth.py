#!/usr/bin/env python3
"""Length of the longest chain of dependent instructions in the links of the cart-pole solver's loop-carried
recurrences, read off the compiled code of scripts/micro/chain_probes.hip (needs hipcc, no GPU).

For every probe kernel the script walks the gfx950 assembly in program order (selects are predicated vector code;
the few scalar branches guard slow paths and are walked through), keeps for every 32-bit register the depth of the
instruction that last wrote it, and reports the depth of the data operands of the final store: depth(instr) =
1 + max(depth(source registers)), loads and waits count 0.  Writes profiles/critical_path.json, which bench.py
turns into the dependency bound of a launch (chain length x measured dependent-issue latency x steps x iterations).

    python scripts/chain_depth.py [--keep-asm FILE]
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "scripts", "micro", "chain_probes.hip")
OUT = os.path.join(ROOT, "profiles", "critical_path.json")

REG = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]|\b(vcc|exec|m0|scc)\b")


def regs_of(operand):
    out = []
    for m in REG.finditer(operand):
        if m.group(1):
            out.append((m.group(1), int(m.group(2))))
        elif m.group(3):
            out.extend((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
        else:
            out.append((m.group(6), 0))
    return out


def analyse(lines):
    """lines: instruction lines of one kernel.  Returns (depth of the last store's data, fp64 ops on that chain,
    instruction count)."""
    depth, f64 = {}, {}
    last_store = (0, 0)
    n_instr = 0
    for ln in lines:
        ln = ln.split(";")[0].strip()
        if not ln or ln.endswith(":") or ln.startswith("."):
            continue
        parts = ln.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if op.startswith(("s_waitcnt", "s_nop", "s_endpgm", "s_branch", "s_cbranch", "s_barrier", "s_sleep")):
            continue
        n_instr += 1
        is_store = "_store_" in op or op.startswith("ds_write")
        is_load = "_load_" in op or op.startswith(("s_load", "ds_read"))
        n_dst = 0 if is_store else 1
        if op.startswith("v_div_scale") or (op.startswith(("v_add_co", "v_sub_co", "v_addc_co", "v_subb_co"))):
            n_dst = 2
        if op.startswith(("s_cmp", "v_cmpx")):
            n_dst = 0
        dsts = [r for o in ops[:n_dst] for r in regs_of(o)]
        srcs = [r for o in ops[n_dst:] for r in regs_of(o)]
        if op.startswith("s_cmp"):
            dsts = [("scc", 0)]
        if op.startswith(("s_cselect", "s_addc", "s_subb")):
            srcs.append(("scc", 0))
        if op.startswith(("v_div_fmas",)) or (op.startswith("v_cndmask") and op.endswith("_e32") and "vcc" not in ln):
            srcs.append(("vcc", 0))
        if op.startswith(("v_fmac", "v_mac")):  # the destination is also the addend
            srcs.extend(dsts)
        d_in = max([depth.get(r, 0) for r in srcs], default=0)
        f_in = max([f64.get(r, 0) for r in srcs if depth.get(r, 0) == d_in], default=0)
        cost = 0 if (is_load or is_store) else 1
        is_f64 = cost and ("_f64" in op)
        d, f = d_in + cost, f_in + (1 if is_f64 else 0)
        if is_store:
            last_store = (d, f)
        for r in dsts:
            depth[r], f64[r] = d, f
    return last_store[0], last_store[1], n_instr


def main():
    keep = sys.argv[sys.argv.index("--keep-asm") + 1] if "--keep-asm" in sys.argv else None
    with tempfile.TemporaryDirectory() as tmp:
        asm = keep or os.path.join(tmp, "probes.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17",
                        "-ffp-contract=off", "-S", "--cuda-device-only", "-o", asm, SRC], check=True)
        text = open(asm).read().splitlines()
    probes = {}
    name, body = None, []
    for ln in text:
        m = re.match(r"^(probe_\w+):", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(ln)
            if "s_endpgm" in ln:
                d, f, n = analyse(body)
                probes[name] = {"chain_instructions": d, "of_which_fp64": f, "instructions_in_kernel": n}
                name = None
    # The recurrences (what must happen one after the other from one time step to the next):
    #  backward sweep, V(i+1) -> V(i)   (ilqr_core.py:127-177, m = 1)
    #    reg = V_xx + mu I (1), f_u^T reg (2 non-zero rows of f_u: 2), Q_uu = l_uu + (.) f_u (3)          6
    #    factor: sqrt, reciprocal                                                            probe_factor
    #    warm start (Q_u r) r                                                                            2
    #    boxQP outcome                                                                        probe_guess
    #    gains -(Q_ux r) r  (1 more op after the control is known: the clamp decides K = 0)              1
    #    V_xx = Q_xx + K^T Q_uu K + K^T Q_ux + Q_ux^T K, symmetrised: 2 + 1 + 2 + 2 + 2                  9
    #  rollout, x(t) -> x(t+1)   (ilqr_core.py:85-107): the longer of the two interlocked cycles
    #    angle:    probe_angle
    #    velocity: probe_velocity (control law + accelerations)
    bw_fixed = 6 + 2 + 1 + 9
    bw = bw_fixed + probes["probe_factor"]["chain_instructions"] + probes["probe_guess"]["chain_instructions"]
    fw = max(probes["probe_angle"]["chain_instructions"], probes["probe_velocity"]["chain_instructions"])
    out = {
        "probes": probes,
        "backward_step_chain": bw, "backward_step_fixed_part": bw_fixed,
        "rollout_step_chain": fw,
        "dependent_issue_cycles": 7.5,
        "shader_clock_hz": 2.4e9,
        "note": "chain lengths in dependent instructions per time step, counted by scripts/chain_depth.py from the "
                "gfx950 code of scripts/micro/chain_probes.hip; 7.5 cycles = measured issue interval of dependent "
                "fp64 instructions of one wavefront (scripts/micro/exec_skip.hip)",
    }
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
