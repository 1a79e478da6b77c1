"""Audit of hipcc output for kernels that hide register loads in inline asm (cdna_hip_programming.md §5.7 item 1):
between an inline-asm `global_load_*` and the next inline-asm `s_waitcnt vmcnt(0)`, no instruction may read or write
the load's destination registers (hipcc considers them written at ASMEND); likewise between an inline-asm `ds_read_*`
and the inline-asm MFMA that consumes it behind its own counted `lgkmcnt`.  Linear scan of the .s text per kernel
(every asm wait in these kernels post-dominates the loads of its pipeline stage, so program order is sufficient).

usage: python tools/audit_asm_loads.py file.s [kernel-name-substring]
"""
import re
import sys


def regs(code):
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', code):
        out.append((int(m.group(1)), int(m.group(2))) if m.group(1) else (int(m.group(3)), int(m.group(3))))
    return out


VM_PREFIXES = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic",
               "scratch_load", "scratch_store", "flat_load", "flat_store")


def audit(lines, name):
    """Linear model of the vector-memory queue: every VM instruction is appended in program order; `s_waitcnt vmcnt(N)`
    (inside or outside asm) retires all but the N youngest.  An asm load with a VGPR destination stays `pending` until
    retired; touching a pending destination is a violation.  Asm LDS reads stay pending until an asm MFMA consumes them."""
    vm, lds_pending, bad, in_asm, n_loads = [], [], 0, False, 0   # vm: list of (dest range or None)
    for ln, line in enumerate(lines, 1):
        t = line.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        code = t.split(';')[0].strip()
        if not code or code.endswith(':') or code.startswith('.'):
            continue
        if code.startswith(VM_PREFIXES):
            if in_asm and code.startswith('global_load') and '_lds_' not in code:   # LDS-DMA has no VGPR destination
                r = regs(code)
                for a, b in r[1:]:   # address registers must not be an in-flight destination
                    for d in vm:
                        if d is not None and a <= d[1] and b >= d[0]:
                            print(f"{name}:{ln}: address reads in-flight destination: {t}")
                            bad += 1
                vm.append(r[0])
                n_loads += 1
            else:
                vm.append(None)
            continue
        m = re.search(r'vmcnt\((\d+)\)', code) if code.startswith('s_waitcnt') else None
        if m:
            keep = int(m.group(1))
            vm = vm[len(vm) - keep:] if keep > 0 else []
            if 'lgkmcnt' not in code and not in_asm:
                continue
        if code.startswith('s_waitcnt'):
            continue
        if in_asm and code.startswith('ds_read'):      # asm LDS read: pending until an asm MFMA consumes it
            lds_pending.append(regs(code)[0])
            n_loads += 1
            continue
        if in_asm and code.startswith('v_mfma'):
            for a, b in regs(code):
                lds_pending[:] = [(lo, hi) for lo, hi in lds_pending if not (a <= hi and b >= lo)]
            continue
        if in_asm:
            continue
        for a, b in regs(code):
            for d in vm:
                if d is not None and a <= d[1] and b >= d[0]:
                    print(f"{name}:{ln}: touches in-flight v[{d[0]}:{d[1]}]: {t}")
                    bad += 1
            for lo, hi in lds_pending:
                if a <= hi and b >= lo:
                    print(f"{name}:{ln}: touches in-flight LDS destination v[{lo}:{hi}]: {t}")
                    bad += 1
    return n_loads, bad


def main():
    text = open(sys.argv[1]).read().split('\n')
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    cur, buf, total_bad = None, [], 0
    for line in text:
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur, buf = m.group(1), []
        elif cur is not None:
            buf.append(line)
            if 's_endpgm' in line:
                if want in cur:
                    n, bad = audit(buf, cur)
                    if n:
                        print(f"{cur}: {n} asm loads, {bad} violations")
                    total_bad += bad
                cur = None
    sys.exit(1 if total_bad else 0)


if __name__ == '__main__':
    main()
