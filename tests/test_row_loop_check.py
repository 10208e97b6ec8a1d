"""The build's guard of the fused pass's row loops (tools/check_row_loops.py: a scratch access, an s_waitcnt vmcnt(0) or a store off the
counted path inside them turns the loop's counted wait into a wait for the row's own stores).  The parser is checked on listings here;
the real disassembly of the library runs in __graft_entry__.build()."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_row_loops", os.path.join(ROOT, "tools", "check_row_loops.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

HEAD = "_ZN6sarpro12_GLOBAL__N_117k_clahe_rgb_fusedENS_12ClaheRgbArgsE: ; @x\n"
GOOD_LOOP = """.LBB5_10:
	global_load_dwordx4 v[2:5], v[2:3], off
	buffer_store_dwordx4 v[50:53], v119, s[68:71], s48 offen nt
	buffer_store_dwordx4 v[56:59], v120, s[68:71], s48 offen nt
	buffer_store_byte v12, v88, s[68:71], s48 offen
	s_waitcnt vmcnt(3)
	s_cbranch_scc1 .LBB5_10
"""


def test_a_clean_row_loop_passes():
    bad, seen = chk.check(HEAD + GOOD_LOOP + ".Lfunc_end5:\n")
    assert seen == 1 and not bad, bad


def test_a_reload_a_full_wait_and_a_stray_store_are_reported():
    for line, what in (("\tscratch_load_dword v4, off, off offset:20\n", "scratch_load"), ("\ts_waitcnt vmcnt(0)\n", "vmcnt(0)"),
                       ("\tglobal_store_byte v[6:7], v21, off\n", "stores on the straight-line path")):
        asm = HEAD + GOOD_LOOP.replace("\ts_waitcnt vmcnt(3)\n", line + "\ts_waitcnt vmcnt(3)\n") + ".Lfunc_end5:\n"
        bad, seen = chk.check(asm)
        assert seen == 1 and len(bad) == 1 and what in bad[0], (what, bad)


def test_other_kernels_and_outer_loops_are_not_looked_at():
    other = "_ZN6sarpro12_GLOBAL__N_112k_compose_u8ILi16ELb0EEEvNS_11ComposeArgsE: ; @y\n" + GOOD_LOOP.replace("vmcnt(3)", "vmcnt(0)") + ".Lfunc_end9:\n"
    outer = HEAD + ".LBB5_1:\n\ts_waitcnt vmcnt(0)\n" + GOOD_LOOP + "\ts_branch .LBB5_1\n.Lfunc_end5:\n"
    for asm in (other, outer):
        bad, seen = chk.check(asm)
        assert not bad, bad
