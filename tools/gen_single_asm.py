#!/usr/bin/env python3
"""Generates aligner_amd/csrc/aln_single_unit.inc: the hand-scheduled gfx950 instruction stream of the steady state of
the single-pair core-local fill (one wave = one strip, lane = R consecutive rows, anti-diagonal skew; aln_fast.h).

Why asm, and why ONE statement for the whole steady loop.  A lone wave issues one instruction per ~4-5 cycles whatever
its kind (VALU, SALU, LDS, s_nop) and nothing hides a memory round trip for it, so the loop is written to the minimum
instruction count with every hazard slot filled by useful work, and it never waits on memory:
  * LDS: the two reads of a step (profile bytes, query offset) are issued two steps before their use (4-deep register
    rotation, s_waitcnt lgkmcnt(2));
  * boundary cells from the strip above arrive as 4-byte granules (the T value, never zero; the row is zeroed before
    every pass).  The loads of the next TWO 16-column groups are always in flight (gA / gB by unit parity).  A hand-off
    costs about a microsecond, and gfx9 may complete loads and stores out of order with respect to each other, so no
    vmcnt count can be trusted here: the destination register is ZEROED before the load is issued and simply read when
    the group is due -- all 16 values non-zero <=> the load has landed AND the producer had published.  Only then
    (rare: start of the strip, or a consumer that caught up) the wave drains, re-loads and polls, bounded, watching
    the abort word.
  * stores (bottom row to the strip below, direction words, bottom-row record of the last strip) are fire-and-forget.
Everything that is in flight is drained before the statement ends, so every output is valid when the compiler sees it.

Per step i (k = ku + i):
  top-in   X' <- lanes 0..3: boundary group G rotated by i (row_ror; only lane 0 matters); lanes 1..63: lane-1's bottom
           cell (wave_shr:1)
  LDS      P[(i+2)%4] <- profile bytes of step i+2 (address = prow + Q[(i+2)%4]);  Q[i%4] <- query offset of step i+4
  cells    penalty select (Beginning above -> del, else ext), three candidate keys, v_max3, T form, Beginning tag,
           2 direction bits, packed end-cell tracker
  bottom   cell -> 64-deep lane shift register towards lane 63's publisher (wave_shl:1)
gfx950 hazards honoured by construction: VALU writes VCC -> VALU reads VCC needs 2 wait states; VALU writes a VGPR -> DPP
reads it needs 2.

Variants: R in {1, 2} x {FIRST (no strip above: G = 2), MID, LAST (no strip below; records its bottom row)}.
Run:  python tools/gen_single_asm.py   (rewrites the .inc; the build does not run it)
"""
import os

SDWA = "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_%d"


def step(i, R, dw, kind="MID", masked=False, sem="LOCAL"):
    X_old, X_new = "%%[X%d]" % (i % 2), "%%[X%d]" % ((i + 1) % 2)
    O_old, O_new = "%%[O%d]" % (i % 2), "%%[O%d]" % ((i + 1) % 2)
    P_cur, P_nxt = "%%[P%d]" % (i % 4), "%%[P%d]" % ((i + 2) % 4)
    Q_adr, Q_nxt = "%%[Q%d]" % ((i + 2) % 4), "%%[Q%d]" % (i % 4)
    ror = "quad_perm:[0,1,2,3]" if i == 0 else "row_ror:%d" % i
    pwread = "ds_read_u8" if R == 1 else "ds_read_u16"
    L = []
    L.append("v_mov_b32_dpp %s, %%[G] %s row_mask:0x1 bank_mask:0x1" % (X_new, ror))
    L.append("v_mov_b32_dpp %s, %%[TL] wave_shr:1 row_mask:0xf bank_mask:0xf" % X_new)
    if sem == "LOCAL":
        L.append("v_cmp_eq_u32 vcc, %%[two], %s" % X_new)
    L.append("WAIT")                                      # for the two reads issued two steps ago (count filled in later)
    L.append("v_add_u32 %%[la], %%[prow], %s" % Q_adr)
    L.append("%s %s, %%[la] ;M" % (pwread, P_nxt))
    L.append("ds_read_u16 %s, %%[qop] offset:%d ;MQ" % (Q_nxt, 2 * (i + 4)))
    if i == 8 and kind != "FIRST":
        # LDS hand-off: the next unit's group is read half a unit ahead (vrin already points at it).  With the global
        # hand-off vrin points at a scratch LDS word and gL is never looked at.
        L.append("ds_read_b32 %[gL], %[vrin] ;M")
    if masked:
        L.append("s_mov_b64 exec, %[em]")                 # only the lanes whose column exists: 0 <= k - lane < N
        L.append("s_or_b64 %[um], %[um], %[em]")          # lanes that had a cell in this unit (for the tracker fold)
    if sem == "GLOBAL":
        # core global (simple/mod.rs:72-98): the carried penalty is ext everywhere past cell (1,1); no Beginning inside the
        # matrix, no end-cell tracker: six instructions per cell
        if R == 1:
            L.append("v_add_u32_sdwa %%[c0], %s, sext(%s) %s" % (X_old, P_cur, SDWA % 0))
            L.append("v_add_u32 %%[ta], %s, %%[ne]" % X_new)
            L.append("v_add3_u32 %[tb], %[TL], %[ne], -1")
            L.append("v_max3_i32 %[k0], %[ta], %[tb], %[c0]")
            L.append("v_and_or_b32 %[TL], %[k0], -4, 2")
            L.append("v_alignbit_b32 %s, %%[k0], %s, 2" % (dw, dw))
            L.append("v_mov_b32 %s, %%[TL]" % O_new)
        else:
            L.append("v_add_u32_sdwa %%[c0], %s, sext(%s) %s" % (X_old, P_cur, SDWA % 0))
            L.append("v_add_u32_sdwa %%[c1], %%[T0], sext(%s) %s" % (P_cur, SDWA % 1))
            L.append("v_add_u32 %%[ta], %s, %%[ne]" % X_new)
            L.append("v_add3_u32 %[tb], %[T0], %[ne], -1")
            L.append("v_max3_i32 %[k0], %[ta], %[tb], %[c0]")
            L.append("v_and_or_b32 %[T0], %[k0], -4, 2")
            L.append("v_alignbit_b32 %s, %%[k0], %s, 2" % (dw, dw))
            L.append("v_add_u32 %[ta], %[T0], %[ne]")
            L.append("v_add3_u32 %[tb], %[TL], %[ne], -1")
            L.append("v_max3_i32 %[c1], %[ta], %[tb], %[c1]")
            L.append("v_and_or_b32 %[TL], %[c1], -4, 2")
            L.append("v_alignbit_b32 %s, %%[c1], %s, 2" % (dw, dw))
            L.append("v_mov_b32 %s, %%[TL]" % O_new)
    elif R == 1:
        L.append("v_add_u32_sdwa %%[c0], %s, sext(%s) %s" % (X_old, P_cur, SDWA % 0))
        L.append("v_cndmask_b32 %[np], %[ne], %[nd], vcc")
        L.append("v_add_u32 %%[ta], %s, %%[np]" % X_new)
        L.append("v_add3_u32 %[tb], %[TL], %[np], -1")
        L.append("v_max3_i32 %[k0], %[ta], %[tb], %[c0]")
        L.append("v_and_or_b32 %[TL], %[k0], -4, 2")
        L.append("v_max_u32 %[k0], %[k0], 3")
        L.append("v_mov_b32 %s, %%[TL]" % O_new)
        if i == 0 and not masked:
            L.append("v_lshl_add_u32 %%[u0], %%[TL], 11, %d" % (15 - i))     # the unit's own tracker starts here
            L.append("v_alignbit_b32 %s, %%[k0], %s, 2" % (dw, dw))
        else:
            L.append("v_lshl_add_u32 %%[p0], %%[TL], 11, %d" % (15 - i))
            L.append("v_alignbit_b32 %s, %%[k0], %s, 2" % (dw, dw))
            L.append("v_max_i32 %[u0], %[u0], %[p0]")
    else:
        # row 1's diagonal is row 0's previous cell, its top is row 0's new cell
        L.append("v_add_u32_sdwa %%[c0], %s, sext(%s) %s" % (X_old, P_cur, SDWA % 0))
        L.append("v_add_u32_sdwa %%[c1], %%[T0], sext(%s) %s" % (P_cur, SDWA % 1))
        L.append("v_cndmask_b32 %[np], %[ne], %[nd], vcc")
        L.append("v_add_u32 %%[ta], %s, %%[np]" % X_new)
        L.append("v_add3_u32 %[tb], %[T0], %[np], -1")
        L.append("v_max3_i32 %[k0], %[ta], %[tb], %[c0]")
        L.append("v_cmp_gt_u32 vcc, 3, %[k0]")
        L.append("v_and_or_b32 %[T0], %[k0], -4, 2")
        L.append("v_max_u32 %[k0], %[k0], 3")
        L.append("v_cndmask_b32 %[np], %[ne], %[nd], vcc")
        L.append("v_add_u32 %[ta], %[T0], %[np]")
        L.append("v_add3_u32 %[tb], %[TL], %[np], -1")
        L.append("v_max3_i32 %[c1], %[ta], %[tb], %[c1]")
        L.append("v_alignbit_b32 %s, %%[k0], %s, 2" % (dw, dw))
        L.append("v_and_or_b32 %[TL], %[c1], -4, 2")
        L.append("v_max_u32 %[c1], %[c1], 3")
        L.append("v_mov_b32 %s, %%[TL]" % O_new)
        if i == 0 and not masked:
            L.append("v_lshl_add_u32 %%[u0], %%[T0], 11, %d" % (15 - i))
            L.append("v_alignbit_b32 %s, %%[c1], %s, 2" % (dw, dw))
            L.append("v_lshl_add_u32 %%[u1], %%[TL], 11, %d" % (15 - i))
        else:
            L.append("v_lshl_add_u32 %%[p0], %%[T0], 11, %d" % (15 - i))
            L.append("v_alignbit_b32 %s, %%[c1], %s, 2" % (dw, dw))
            L.append("v_max_i32 %[u0], %[u0], %[p0]")
            L.append("v_lshl_add_u32 %%[p0], %%[TL], 11, %d" % (15 - i))
            L.append("v_max_i32 %[u1], %[u1], %[p0]")
    if masked:
        # next step's lanes: everything moves up one lane, lane 0 stays in while columns remain (krem = N - 1 - k)
        L.append("s_mov_b64 exec, -1")
        L.append("s_lshl_b64 %[em], %[em], 1")
        L.append("s_cmp_gt_i32 %[krem], 0")
        L.append("s_cselect_b64 %[et], 1, 0")
        L.append("s_or_b64 %[em], %[em], %[et]")
        L.append("s_sub_u32 %[krem], %[krem], 1")
        L.append("v_mov_b32_dpp %s, %s wave_shl:1 row_mask:0xf bank_mask:0xf" % (O_new, O_old))
        return L
    L.append("v_mov_b32_dpp %s, %s wave_shl:1 row_mask:0xf bank_mask:0xf" % (O_new, O_old))
    return L


def poll_tail(u, uid, tag, again):
    """s_sleep, count, every 1024 polls look at the abort word, give up after 2^22 polls."""
    L = []
    L.append("s_sleep 1")
    L.append("s_add_u32 %[spin], %[spin], 1")
    L.append("s_and_b32 %[st], %[spin], 0x3ff")
    L.append("s_cmp_lg_u32 %[st], 0")
    L.append("s_cbranch_scc1 %s" % again)
    L.append("global_load_dword %[G], %[vzero], %[abortp] sc1")
    L.append("s_waitcnt vmcnt(0)")
    L.append("v_readfirstlane_b32 %[st], %[G]")
    L.append("s_cmp_lg_u32 %[st], 0")
    L.append("s_cbranch_scc1 Labort_%s" % uid)
    L.append("s_cmp_lt_u32 %[spin], 0x400000")
    L.append("s_cbranch_scc1 %s" % again)
    L.append("s_branch Labort_%s" % uid)
    return L


def acquire(u, uid, masked=False):
    """Group of unit u -> G.  amode 0: granules in HBM/L2 (register gA / gB by parity; the group two units on is
    requested into the same register).  amode 1: the producer is a wave of this workgroup and the group sits in an LDS
    ring (read half a unit ahead into gL; consumed slots are zeroed, which is what lets the producer reuse them)."""
    g = "%[gA]" if u % 2 == 0 else "%[gB]"
    L = []

    def ready_test():
        T = ["v_cmp_ne_u32 vcc, 0, %[G]"]
        if masked:
            T.append("s_or_b64 vcc, vcc, %[et]")         # columns >= N are never published
        T.append("s_cmp_eq_u64 vcc, exec")
        return T

    if masked:
        L.append("v_cmp_le_u32 %[et], %[n4], %[vsrc]")   # et: lanes whose column of this group is >= N (vsrc = 4 * column)
    L.append("s_cmp_eq_u32 %[amode], 0")
    L.append("s_cbranch_scc0 Lacql%d_%s" % (u, uid))
    # ---------------- global hand-off
    L.append("v_mov_b32 %%[G], %s" % g)
    L += ready_test()
    L.append("s_cbranch_scc1 Lrdy%d_%s" % (u, uid))
    L.append("s_mov_b32 %[spin], 0")
    L.append("Lpoll%d_%s:" % (u, uid))
    L.append("s_waitcnt vmcnt(0)")                     # a load that was merely late
    L.append("v_mov_b32 %%[G], %s" % g)
    L += ready_test()
    L.append("s_cbranch_scc1 Lrdy%d_%s" % (u, uid))
    L.append("global_load_dword %s, %%[vsrc], %%[gin] sc1" % g)
    L += poll_tail(u, uid, "g", "Lpoll%d_%s" % (u, uid))
    L.append("Lrdy%d_%s:" % (u, uid))
    L.append("v_mov_b32 %s, 0" % g)
    L.append("global_load_dword %s, %%[vsrc], %%[gin] offset:128 sc1" % g)
    L.append("s_branch Lgo%d_%s" % (u, uid))
    # ---------------- LDS hand-off
    L.append("Lacql%d_%s:" % (u, uid))
    L.append("v_mov_b32 %[G], %[gL]")
    L += ready_test()
    L.append("s_cbranch_scc1 Lgo%d_%s" % (u, uid))
    L.append("s_mov_b32 %[spin], 0")
    L.append("Lpolll%d_%s:" % (u, uid))
    L.append("ds_read_b32 %[G], %[vrin]")
    L.append("s_waitcnt lgkmcnt(0)")
    L += ready_test()
    L.append("s_cbranch_scc1 Lgo%d_%s" % (u, uid))
    L += poll_tail(u, uid, "l", "Lpolll%d_%s" % (u, uid))
    # ---------------- both: the LDS traffic of the main line is the same whatever the mode (the lgkmcnt counts of the
    # steps depend on it): the consumed ring slots are zeroed -- which is what lets the producer reuse them -- or, with
    # the global hand-off, a scratch word is (astep = 0 keeps vrin on it)
    L.append("Lgo%d_%s:" % (u, uid))
    L.append("ds_write_b32 %[vrin], %[vzero] ;M")
    L.append("v_add_u32 %[vrin], %[astep], %[vrin]")
    L.append("v_and_or_b32 %[vrin], %[vrin], %[vrmask], %[vrbin]")
    L.append("v_add_u32 %[vsrc], 64, %[vsrc]")
    return L


def publish(u, uid, masked=False):
    """Bottom cells of columns ku-63 .. ku-48 (lanes 48..63 of the shift register) to the strip below.  pmode 0: one
    write-through store to the granule row.  pmode 1: the consumer is a wave of this workgroup: LDS ring; the slots
    must have been consumed (zero) -- they were read at the start of the unit (chk), so this costs no LDS round trip.
    The ds_write is issued in both modes (pmode 0: to scratch words) so that the LDS traffic of the main line is fixed."""
    L = []
    mask = "%[m48]"
    if masked:
        L.append("v_cmp_le_i32 %[et], 0, %[vpub]")       # ramp-up: lanes 48..63 still hold columns < 0 (vpub = 4 * column)
        L.append("s_and_b64 %[et], %[et], %[m48]")
        mask = "%[et]"
    L.append("s_mov_b64 exec, %s" % mask)
    L.append("s_cmp_eq_u32 %[pmode], 0")
    L.append("s_cbranch_scc0 Lpubl%d_%s" % (u, uid))
    L.append("global_store_dword %[vpub], %[O0], %[gout] sc1")
    L.append("s_branch Lpubw%d_%s" % (u, uid))
    L.append("Lpubl%d_%s:" % (u, uid))
    L.append("v_cmp_ne_u32 vcc, 0, %[chk]")
    L.append("s_cmp_eq_u64 vcc, 0")
    L.append("s_cbranch_scc1 Lpubw%d_%s" % (u, uid))
    # ring full (the consumer is a whole lap behind): poll until the slots are free
    L.append("s_mov_b32 %[spin], 0")
    L.append("Lpubp%d_%s:" % (u, uid))
    L.append("ds_read_b32 %[chk], %[vrout]")
    L.append("s_waitcnt lgkmcnt(0)")
    L.append("v_cmp_ne_u32 vcc, 0, %[chk]")
    L.append("s_cmp_eq_u64 vcc, 0")
    L.append("s_cbranch_scc1 Lpubw%d_%s" % (u, uid))
    L.append("s_sleep 1")
    L.append("s_add_u32 %[spin], %[spin], 1")
    L.append("s_cmp_lt_u32 %[spin], 0x400000")
    L.append("s_cbranch_scc1 Lpubp%d_%s" % (u, uid))
    L.append("s_mov_b64 exec, -1")
    L.append("s_branch Labort_%s" % uid)
    L.append("Lpubw%d_%s:" % (u, uid))
    L.append("ds_write_b32 %[vrout], %[O0] ;M")
    L.append("s_mov_b64 exec, -1")
    L.append("v_add_u32 %[vpub], 64, %[vpub]")
    L.append("v_add_u32 %[vrout], %[pstep], %[vrout]")
    L.append("v_and_or_b32 %[vrout], %[vrout], %[vrmask], %[vrbout]")
    return L


def loop(R, kind, masked=False, sem="LOCAL"):
    U = 4 // R                      # units per quad
    uid = "%="
    L = []
    L.append("Lloop_%s:" % uid)
    for u in range(U):
        if kind != "FIRST":
            L += acquire(u, uid, masked)
        if kind != "LAST":
            L.append("ds_read_b32 %[chk], %[vrout] ;M")      # LDS publish: are this unit's ring slots free? (looked at below)
        if masked and sem == "LOCAL":
            L.append("v_mov_b32 %[u0], 0x80000000")          # a lane may join in mid-unit: no cell yet = INT_MIN
            if R == 2:
                L.append("v_mov_b32 %[u1], 0x80000000")
        if masked:
            L.append("s_mov_b64 %[um], 0")
        for i in range(16):
            if R == 1:
                dw = "%%[w%d]" % u
            else:
                dw = "%%[w%d]" % (2 * u + (0 if i < 8 else 1))
            L += step(i, R, dw, kind, masked, sem)
        L.append("v_add_u32 %[qop], 32, %[qop]")
        # end-cell tracker: within the unit the packs carry 15 - i (an inline constant); the unit's term -- 16 * (units
        # left in the 2048-step chunk), kt = 2047 - (k & 2047) = that + 15 - i -- is added once per unit and row
        if sem == "LOCAL":
            if masked:
                L.append("s_mov_b64 exec, %[um]")            # lanes that had a cell in this unit
            L.append("v_add_u32 %[p0], %[u0], %[kt]")
            L.append("v_max_i32 %[r0], %[r0], %[p0]")
            if R == 2:
                L.append("v_add_u32 %[p0], %[u1], %[kt]")
                L.append("v_max_i32 %[r1], %[r1], %[p0]")
            if masked:
                L.append("s_mov_b64 exec, -1")
            L.append("s_sub_u32 %[kt], %[kt], 16")
        elif kind == "FIRST":
            # core global, strip 0: the row above is the border H[0][x] = -x del (simple/mod.rs:59-62); G holds it for the
            # unit's 16 columns and moves on by 16 columns' worth (kt = -64 del in T units)
            L.append("v_add_u32 %[G], %[kt], %[G]")
        if kind != "LAST":
            L += publish(u, uid, masked)
        else:
            # the lane that owns row M records its direction words (tag 3 <=> H == 0), one per block
            L.append("s_mov_b64 exec, %[zmask]")
            if R == 1:
                L.append("global_store_dword %%[vz], %%[w%d], %%[zbase]" % u)
            else:
                L.append("global_store_dword %%[vz], %%[w%d], %%[zbase]" % (2 * u))
                L.append("global_store_dword %%[vz], %%[w%d], %%[zbase] offset:4" % (2 * u + 1))
            L.append("s_mov_b64 exec, -1")
            L.append("v_add_u32 %%[vz], %d, %%[vz]" % (4 * R))
    # the quad's direction words: 16 bytes per lane, 1 KiB per wave
    L.append("s_cmp_eq_u32 %[sdirs], 0")
    L.append("s_cbranch_scc1 Lnodir_%s" % uid)
    for j in range(4):
        L.append("global_store_dword %%[vdir], %%[w%d], %%[dbase] offset:%d" % (j, 4 * j))
    L.append("Lnodir_%s:" % uid)
    L.append("v_add_u32 %[vdir], 0x400, %[vdir]")
    L.append("s_add_u32 %%[ku], %%[ku], %d" % (16 * U))
    L.append("s_cmp_lt_u32 %[ku], %[kend]")
    L.append("s_cbranch_scc1 Lloop_%s" % uid)
    L.append("s_mov_b32 %[st], 0")
    L.append("s_branch Lexit_%s" % uid)
    L.append("Labort_%s:" % uid)
    L.append("s_mov_b32 %[st], 1")
    L.append("Lexit_%s:" % uid)
    L.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return fill_waits(L)


def fill_waits(L):
    """LDS operations complete in issue order, so `s_waitcnt lgkmcnt(n)` with n = the number of main-line LDS operations
    issued after the one that is needed waits for exactly that one.  The main line is the loop body in steady state
    (poll paths drain everything and only make a later wait trivially true).  A step's WAIT needs the query-offset read
    (;MQ, issued after the profile read) of two steps before, across units and across the loop's back edge."""
    main = [(j, ln) for j, ln in enumerate(L) if ln == "WAIT" or ";M" in ln]
    n = len(main)
    out = list(L)
    for pos, (j, ln) in enumerate(main):
        if ln != "WAIT":
            continue
        # walk back (cyclically) to the second ;MQ before this wait
        seen_q, younger, p = 0, 0, pos
        while True:
            p = (p - 1) % n
            t = main[p][1]
            if t == "WAIT":
                continue
            if t.endswith(";MQ"):
                seen_q += 1
                if seen_q == 2:
                    break
            younger += 1
        out[j] = "s_waitcnt lgkmcnt(%d)" % younger
    return [ln.replace(" ;MQ", "").replace(" ;M", "") for ln in out]


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "aligner_amd", "csrc", "aln_single_unit.inc")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_single_asm.py -- do not edit.  Steady-state loop of the single-pair core-local fill.\n")
        for R in (1, 2):
            for kind in ("FIRST", "MID", "LAST"):
                lines = loop(R, kind)
                n = sum(1 for ln in lines if not ln.endswith(":"))
                f.write("// R = %d, %s strip: %d instructions per quad of %d steps\n" % (R, kind, n, 64 // R))
                f.write("#define ALN_STEADY_ASM_R%d_%s \\\n" % (R, kind))
                for j, ln in enumerate(lines):
                    last = j + 1 == len(lines)
                    f.write('    "%s%s"%s\n' % (ln, "" if last else "\\n\\t", "" if last else " \\"))
                f.write("\n")
                if kind != "FIRST":
                    # the same loop with the cell update under exec = {lanes whose column exists}: the strip's first 64
                    # steps and its tail (ramp-down + what is left of N after the last full quad).  A strip finishes
                    # `lag` steps after the strip above it, so the tail is on the critical path of EVERY strip.
                    lines = loop(R, kind, masked=True)
                    f.write("// R = %d, %s strip: masked quads (ramp-up / tail)\n" % (R, kind))
                    f.write("#define ALN_MASKED_ASM_R%d_%s \\\n" % (R, kind))
                    for j, ln in enumerate(lines):
                        last = j + 1 == len(lines)
                        f.write('    "%s%s"%s\n' % (ln, "" if last else "\\n\\t", "" if last else " \\"))
                    f.write("\n")
        # core global: same loops, six-instruction cells
        for R in (1, 2):
            for kind in ("FIRST", "MID", "LAST"):
                for masked in ((False,) if kind == "FIRST" else (False, True)):
                    lines = loop(R, kind, masked, sem="GLOBAL")
                    name = "ALN_G%s_ASM_R%d_%s" % ("MASKED" if masked else "STEADY", R, kind)
                    f.write("// core global, R = %d, %s strip%s\n" % (R, kind, ", masked quads" if masked else ""))
                    f.write("#define %s \\\n" % name)
                    for j, ln in enumerate(lines):
                        last = j + 1 == len(lines)
                        f.write('    "%s%s"%s\n' % (ln, "" if last else "\\n\\t", "" if last else " \\"))
                    f.write("\n")
    print("wrote", os.path.normpath(out))


if __name__ == "__main__":
    main()
