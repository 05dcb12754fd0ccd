#!/usr/bin/env python3
"""Scan device ISA (hipcc --cuda-device-only -S) for VGPR spill code that sits AHEAD of a block's EXEC restore.

Why (round 6, the `used_vg` discrepancy of the lane-combined wg_reduce at 1 024 threads): in that build the compiler placed
`scratch_store_dwordx2 ... ; 8-byte Folded Spill` at the top of a control-flow JOIN block, before the block's `s_or_b64 exec, exec, s[..]` -- so the
spill ran under the narrowed EXEC of the divergent region that had just ended, the lanes that had skipped the region never saved their value, and the
reload 36 000 instructions later (under full EXEC) handed them whatever the scratch slot held.  The value was the per-thread byte offset of the final
volume-group sum: placements, cpu and memory were right, `used_vg` was not, deterministically, at the one workgroup size whose 128-VGPR budget spills.
The same source in another translation unit allocates registers differently and is correct -- nothing in the source is wrong, and nothing in the source
can rule it out.  Hence this scan: every product kernel, every build.

usage: python profiles/spill_scan.py <file.s> ...      exit code 1 when a suspicious spill / reload is found
A finding = a `Folded Spill` STORE of a VGPR in the prologue of a label-delimited block -- nothing but scalar moves, waits, lane reads and other
spill code between the label and it -- that is followed, still in that prologue, by the block's `s_or_b64 exec, exec, ...`.  (Reloads ahead of a restore
are listed with --reloads: most are legitimate -- the value is consumed inside the narrowed region the restore ends.)"""
import re
import sys

LABEL = re.compile(r"^(\.LBB\d+_\d+|[A-Za-z_][\w$.]*):")
SPILL = re.compile(r"^\s*scratch_(store|load)_\w+\s.*;\s*\d+-byte Folded (Spill|Reload)")
PROLOGUE = re.compile(r"^\s*(s_mov_b\d+|s_waitcnt|s_nop|v_readlane_b32|v_writelane_b32|;|$)")
RESTORE = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,")
BRANCH = re.compile(r"^\s*s_(cbranch_\w+|branch|endpgm|setpc_b64|swappc_b64)\b")


def scan(path, reloads=False):
    findings, kernel, block, pending = [], "?", "?", []
    n_spill = 0
    prologue = False
    with open(path, errors="replace") as f:
        for ln, line in enumerate(f, 1):
            m = LABEL.match(line)
            if m:
                if not m.group(1).startswith(".L"):
                    kernel = m.group(1)
                block, pending, prologue = m.group(1), [], True
                continue
            sp = SPILL.match(line)
            if sp:
                n_spill += 1
                if (sp.group(1) == "store" and prologue) or (reloads and sp.group(1) == "load"):
                    pending.append((ln, line.strip()))
            elif RESTORE.match(line):
                for p in pending:
                    findings.append((kernel, block, p[0], p[1], ln))
                pending = []
            elif BRANCH.match(line):
                pending = []
            elif not PROLOGUE.match(line):
                prologue = False
                if not reloads:
                    pending = []
    return n_spill, findings


def main():
    bad = 0
    reloads = "--reloads" in sys.argv
    for path in [a for a in sys.argv[1:] if not a.startswith("--")]:
        n_spill, findings = scan(path, reloads)
        print(f"{path}: {n_spill} VGPR spill / reload instructions, {len(findings)} ahead of an EXEC restore in their block")
        for kernel, block, ln, text, ln2 in findings:
            print(f"   {kernel} {block} line {ln}: {text}   (exec restored at line {ln2})")
        bad += len(findings)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
