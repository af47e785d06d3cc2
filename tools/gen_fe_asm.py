#!/usr/bin/env python3
"""Generates decaf377_amd/csrc/fe_asm.inc: the gfx950 instruction streams of the Fq multiplier and
squarer (9 x 29-bit limbs, Montgomery R = 2^261, product scanning with the reduction interleaved)
as inline-asm bodies for fq29.hpp.

Why hand-written: measured on MI355X (tools/valu_mix.hip, profiles/r02_valu_mix*.txt) every VALU
instruction of a MAC-dominated stream costs one ~4-cycle issue slot, so the only lever is the
instruction count.  hipcc's build of the C++ multiplier needs 212 VALU instructions per product
(153 v_mad_u64_u32 + a 64-bit add and a v_mov for each m*q0, separate negate and mask) and pads every
dependent MAC with an s_nop (the empty-asm accumulator pin is treated as a dst-forwarding hazard).
These streams are 196 (multiplication) and 168 (squaring) instructions, with no pad:

  column k < 9:   MACs a_i*b_(k-i), m_i*q_(k-i)         v_mad_u64_u32 acc, a, b, acc   (q_j in SGPRs)
                  m_k = -acc mod 2^32                    v_sub_u32 m, 0, acc.lo
                  acc += m_k * q_0   (q_0 = 1)           v_mad_u64_u32 acc, m, 1, acc   (low 32 bits -> 0)
                  acc >>= 29                             v_lshrrev_b64
  column k >= 9:  MACs, r_(k-9) = acc & (2^29-1), acc >>= 29;   the last shift writes the top limb.

The Montgomery digit is taken mod 2^32 instead of mod 2^29 (q = 1 mod 2^47, so -q^-1 = -1 for either
width): it still clears the low 29 bits, saves the mask, and only loosens the output bound from
a*b/R + q to a*b/R + 8q (digits up to 2^32).  The "strict" variants keep the 29-bit digit (one more
v_and_b32 per column) for the places that need the tighter bound (canonicalisation, hash keys).

The accumulator lives in v[2:3] by name (inline-asm operands have no sub-register syntax and the
stream needs acc.lo on its own), declared as a clobber.  m_k shares the register of result limb r_k
(m_k dies in column k+8, r_k is born in column k+9).

Dev tool, run by hand:  python tools/gen_fe_asm.py
"""
import os

NL, RB = 9, 29
MASK = (1 << RB) - 1
QL = [0x00000001, 0x108c0000, 0x00000042, 0x14edfda0, 0x1b00159a, 0x068f2e1b, 0x155982d1, 0x0bd34594, 0x0012ab65]
ACC, ACC_LO = "v[2:3]", "v2"


class Stream:
    def __init__(self):
        self.lines = []
        self.started = False          # has the accumulator been written yet?

    def emit(self, s):
        self.lines.append(s)

    def mac(self, x, y, dst=ACC):
        src2 = ACC if self.started else "0"
        self.emit("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (dst, x, y, src2))
        self.started = True

    def text(self):
        return "\\n\\t".join(self.lines)

    def count(self):
        return len(self.lines)


def reduce_tail(s, m_reg, strict):
    """end of a column k < 9: Montgomery digit, fold m*q0, shift"""
    s.emit("v_sub_u32 %s, 0, %s" % (m_reg, ACC_LO))
    if strict:
        s.emit("v_and_b32 %s, 0x%x, %s" % (m_reg, MASK, m_reg))
    s.emit("v_mad_u64_u32 %s, vcc, %s, 1, %s" % (ACC, m_reg, ACC))
    s.emit("v_lshrrev_b64 %s, %d, %s" % (ACC, RB, ACC))


def result_tail(s, r_reg, last, top):
    s.emit("v_and_b32 %s, 0x%x, %s" % (r_reg, MASK, ACC_LO))
    s.emit("v_lshrrev_b64 %s, %d, %s" % (top if last else ACC, RB, ACC))


def gen(kind, strict):
    """kind: 'mul', 'sqr', 'sqr2x' (2*a^2).  Returns (asm text, operand map, n_instructions)."""
    # operand numbering: outputs r0..r7 (%0-%7, double as m0..m7), top (%8, 64-bit: its low half is
    # limb 8), m8 (%9); then temporaries (doubled limbs), then inputs, then q1..q8 in SGPRs
    r = ["%%%d" % i for i in range(8)]
    top = "%8"
    m = r + ["%9"]
    nxt = 10
    if kind == "mul":
        a = ["%%%d" % (nxt + i) for i in range(NL)]
        nxt += NL
        b = ["%%%d" % (nxt + i) for i in range(NL)]
        nxt += NL
        ntmp = 0
    else:
        ntmp = NL if kind == "sqr2x" else NL - 1
        a2 = ["%%%d" % (nxt + i) for i in range(ntmp)]
        nxt += ntmp
        a = ["%%%d" % (nxt + i) for i in range(NL)]
        nxt += NL
    q = [None] + ["%%%d" % (nxt + i) for i in range(NL - 1)]
    s = Stream()
    if kind != "mul":
        for i in range(ntmp):
            s.emit("v_lshlrev_b32 %s, 1, %s" % (a2[i], a[i]))
    nmac = 0
    for k in range(2 * NL - 1):
        lo, hi = max(0, k - (NL - 1)), min(k, NL - 1)
        if kind == "mul":
            for i in range(lo, hi + 1):
                s.mac(a[i], b[k - i]); nmac += 1
        elif kind == "sqr":
            for i in range(lo, hi + 1):
                if 2 * i < k:
                    s.mac(a2[i], a[k - i]); nmac += 1
            if k % 2 == 0:
                s.mac(a[k // 2], a[k // 2]); nmac += 1
        else:   # 2*a^2: off-diagonal terms 4 a_i a_j = (2a_i)(2a_j), diagonal 2 a_i^2 = (2a_i) a_i
            for i in range(lo, hi + 1):
                if 2 * i < k:
                    s.mac(a2[i], a2[k - i]); nmac += 1
            if k % 2 == 0:
                s.mac(a2[k // 2], a[k // 2]); nmac += 1
        # reduction terms m_i * q_(k-i), q index 1..8
        for i in range(max(0, k - (NL - 1)), min(k - 1, NL - 1) + 1):
            s.mac(m[i], q[k - i]); nmac += 1
        if k < NL:
            reduce_tail(s, m[k], strict)
        else:
            result_tail(s, r[k - NL], k == 2 * NL - 2, top)
    return s.text(), nmac, s.count(), ntmp


def main():
    out = ["// fe_asm.inc -- GENERATED by tools/gen_fe_asm.py; do not edit.  See that script for the scheme.\n"]
    summary = []
    for kind in ("mul", "sqr", "sqr2x"):
        for strict in (False, True):
            if kind == "sqr2x" and strict:
                continue
            text, nmac, n, ntmp = gen(kind, strict)
            name = "D377_ASM_%s%s" % (kind.upper(), "_STRICT" if strict else "")
            out.append("// %s: %d instructions (%d v_mad_u64_u32)\n#define %s \"%s\"\n" % (name, n, nmac, name, text))
            summary.append((name, n, nmac))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "decaf377_amd", "csrc", "fe_asm.inc")
    with open(path, "w") as f:
        f.write("".join(out))
    for s in summary:
        print("%-24s %3d instructions, %3d MACs" % s)
    # column bound of the relaxed (32-bit digit) variant: 9 La Lb + (2^32 - 1)(q1 + .. + q8) + 2^32 + carry < 2^64
    sq = sum(QL[1:])
    room = (1 << 64) - 1 - ((1 << 32) - 1) * sq - (1 << 32) - (1 << 36)
    print("relaxed: La*Lb <= 2^%.3f ; strict: La*Lb <= 2^%.3f" % (
        __import__("math").log2(room / 9), __import__("math").log2(((1 << 64) - 1 - MASK * sq - (1 << 29) - (1 << 36)) / 9)))


if __name__ == "__main__":
    main()
