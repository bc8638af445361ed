// Word offsets of the fields of the LDS scratch block in the device layout (for tools/diag_poison_bisect.py):
//   hipcc --offload-arch=gfx950 tools/scratch_offsets.hip -o /tmp/scratch_offsets && /tmp/scratch_offsets
#include <cstdio>
#include <cstddef>
#include "../robovat_amd/csrc/rv_dev_env.h"
using namespace rv;
#define F(f) printf("%5zu %5zu  %s\n", offsetof(Scratch, f) / 4, (offsetof(Scratch, f) + sizeof(((Scratch*)0)->f)) / 4, #f);
int main() {
  F(frot) F(fv) F(fw) F(axis) F(colv) F(colc) F(colr) F(colmin) F(colmax) F(arm_moving) F(colflag) F(rot) F(iinv) F(tablev) F(groundv)
  F(u) F(wp) F(gstart) F(start_pos) F(start_yaw) F(poses) F(num_waypoints) F(interrupt) F(has_budget) F(max_phase_steps)
  F(loop_break) F(wus_steps) F(wus_stable) F(valid) F(wake) F(bud_sub) F(bud_sub0) F(suspended) F(wus_resume) F(bud_clk) F(bud_t0)
  F(ready) F(res) F(mot) F(sync) F(ik_q) F(ik_J) F(ik_need) F(ik_conv) F(lq) F(vdraw) F(ratio) F(fing_dv) F(fing_vt) F(fing_qd0)
  F(limb_dv) F(limb_vt) F(limb_qd0) F(lcom) F(lA) F(lGq) F(llo) F(lhi) F(ltgt) F(lJa) F(lMiJ) F(linvk) F(jmoving) F(jchg) F(rvec)
  F(jt_applied) F(atflag) F(kin_fresh) F(ftravel) F(far_valid) F(far_n) F(far) F(nearf) F(bnear) F(near_any) F(sep) F(coltravel)
  F(cdelta) F(clr_t) F(clr_b) F(clr_valid) F(jtravel) F(ccoef) F(crun) F(fused_n) F(fused_pending) F(coast_unsafe) F(jlen) F(colext)
  F(fext) F(fmot) F(any_on) F(pairs) F(rf_dist) F(rf_rm) F(cn) F(ow_run) F(olist) F(n_olist) F(wvneed) F(rng)
  printf("%5zu words\n", sizeof(Scratch) / 4);
}
