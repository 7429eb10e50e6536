#!/usr/bin/env python3
"""Lint of the hand-written gfx950 assembly blocks (csrc/vrt_traverse.h and whatever else vrt_device.hip includes), on the
PREPROCESSED device source, so that every macro-built variant of a block (the counting twins, the prefetching form ...) is seen
as the compiler sees it.  The compiler keeps its own values out of the registers a block declares clobbered and out of its
output operands; what nobody checked until round 4 is the other direction -- that a block only WRITES what it declared:

  1. every register named literally in a block (v48, s[68:69], vcc ...) is in the block's clobber list;
  2. every operand a block writes (first operand of an instruction that has a destination) is an OUTPUT operand ("=v", "+v",
     "=s", "+s"), never an input;
  3. no block names a register above the budget the kernels are built for (v0 - v63 for the 8-wave kernels would collide with
     pinned v48 - v57 only if the compiler ran out: the pinned range must stay inside 0..71).

`make -C voxel-raytracing_amd/csrc resources` and tests/test_kernel_resources.py run it next to the register budgets."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "voxel-raytracing_amd", "csrc")
# mnemonics (prefixes) whose FIRST operand is not a destination register
NO_DEST = ("s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_cmp", "s_bitcmp", "s_endpgm", "s_barrier", "s_setprio", "s_sleep",
           "global_store", "flat_store", "scratch_store", "buffer_store", "ds_write", "ds_add", "ds_min", "ds_max", "ds_or", "ds_and",
           "v_cmpx", ".p2align", ".fill", "s_setpc", "s_getpc")
# v_cmp_* writes vcc (e32) or its first operand (e64: an SGPR pair)


def preprocessed():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    p = subprocess.run([hipcc, "--offload-arch=gfx950", "--cuda-device-only", "-std=c++17", "-E", "-P", "vrt_device.hip"],
                       cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0:
        raise RuntimeError("hipcc -E failed:\n" + p.stderr[-3000:])
    return p.stdout


def asm_statements(src):
    """(template text, outputs, inputs, clobbers) of every asm statement with operands"""
    out = []
    for m in re.finditer(r"\basm\s+(?:volatile\s*)?\(", src):
        i = m.end()
        depth, j, in_str = 1, i, False
        while depth and j < len(src):
            ch = src[j]
            if in_str:
                if ch == "\\": j += 1
                elif ch == '"': in_str = False
            elif ch == '"': in_str = True
            elif ch == "(": depth += 1
            elif ch == ")": depth -= 1
            j += 1
        body = src[i:j - 1]
        # split at top-level ':' (outside strings and parentheses)
        parts, cur, depth, in_str, k = [], [], 0, False, 0
        while k < len(body):
            ch = body[k]
            if in_str:
                cur.append(ch)
                if ch == "\\": cur.append(body[k + 1]); k += 1
                elif ch == '"': in_str = False
            elif ch == '"': in_str = True; cur.append(ch)
            elif ch in "([": depth += 1; cur.append(ch)
            elif ch in ")]": depth -= 1; cur.append(ch)
            elif ch == ":" and depth == 0 and not (k + 1 < len(body) and body[k + 1] == ":") and not (k and body[k - 1] == ":"):
                parts.append("".join(cur)); cur = []
            else: cur.append(ch)
            k += 1
        parts.append("".join(cur))
        text = "".join(bytes(s, "utf-8").decode("unicode_escape") for s in re.findall(r'"((?:[^"\\]|\\.)*)"', parts[0]))
        if len(parts) < 2:
            continue
        outs = re.findall(r'\[(\w+)\]\s*"([^"]*)"', parts[1]) if len(parts) > 1 else []
        ins = re.findall(r'\[(\w+)\]\s*"([^"]*)"', parts[2]) if len(parts) > 2 else []
        clob = re.findall(r'"([^"]*)"', parts[3]) if len(parts) > 3 else []
        out.append((text, dict(outs), dict(ins), set(clob)))
    return out


def regs_in(tok):
    """literal registers named by an operand token: v48 -> {v48}; s[68:69] -> {s68, s69}; vcc -> {vcc}"""
    tok = tok.strip()
    m = re.fullmatch(r"([vs])(\d+)", tok)
    if m: return {m.group(1) + m.group(2)}
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", tok)
    if m: return {m.group(1) + str(q) for q in range(int(m.group(2)), int(m.group(3)) + 1)}
    if tok in ("vcc", "vcc_lo", "vcc_hi"): return {"vcc"}
    return set()


def lint(stmts):
    bad, blocks = [], 0
    for text, outs, ins, clob in stmts:
        lines = [l.strip() for l in text.replace("\t", "\n").split("\n")]
        lines = [l for l in lines if l and not l.endswith(":")]
        if len(lines) < 8:
            continue                                            # one-liners (v_mul_legacy ...): operands only
        blocks += 1
        first = lines[1][:40] if len(lines) > 1 else ""
        for l in lines:
            m = re.match(r"([\w.]+)\s*(.*)", l)
            if not m:
                continue
            mn, ops = m.group(1), [o.strip() for o in re.split(r",(?![^\[]*\])", m.group(2))] if m.group(2) else []
            # 1. literal registers anywhere in the line
            for o in ops:
                for tokn in re.findall(r"[vs]\[\d+:\d+\]|\b[vs]\d+\b|\bvcc\b", o):
                    for r in regs_in(tokn):
                        if r not in clob:
                            bad.append(f"block '{first}...': `{l}` names {r}, which the block does not declare clobbered")
                        if r[0] == "v" and r != "vcc" and int(r[1:]) > 71:
                            bad.append(f"block '{first}...': `{l}` pins {r} above the 72-VGPR budget of the megakernels")
            # 2. the destination
            if not ops or mn.startswith(NO_DEST):
                continue
            dest = ops[0]
            if mn.startswith("v_cmp") and mn.endswith("_e32"):
                continue                                        # writes vcc implicitly (checked as a literal above when named)
            mo = re.fullmatch(r"%\[(\w+)\]", dest)
            if mo and mo.group(1) not in outs:
                bad.append(f"block '{first}...': `{l}` writes %[{mo.group(1)}], an INPUT operand ({ins.get(mo.group(1), '?')})")
            if mo and mo.group(1) in outs and not outs[mo.group(1)].startswith(("+", "=")):
                bad.append(f"block '{first}...': `{l}` writes %[{mo.group(1)}] whose constraint is {outs[mo.group(1)]}")
    return bad, blocks


if __name__ == "__main__":
    bad, blocks = lint(asm_statements(preprocessed()))
    print(f"{blocks} assembly blocks checked")
    if bad:
        print("\n".join(sorted(set(bad))), file=sys.stderr)
        sys.exit(1)
    print("every block writes only what it declares")
