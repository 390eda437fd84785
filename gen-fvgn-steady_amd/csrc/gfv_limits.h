// Dispatch limits of gfv_rowtile_chain / gfv_trans_mlp_*: which kernel family takes a launch of M rows.  ONE table (round 6; the
// families used to call getenv 2 - 4 times on every eager launch): a limit is read from its environment variable ONCE, at first use,
// and can be moved afterwards with gfv_set_limit (include/gfv.h) - the tests' and the A/B tools' handle.
#pragma once
enum {
  GFV_LIM_CBWD_ON = 0,        // GFV_CBWD            1      the column-owner small-tile backward (cbwd.hip)
  GFV_LIM_CBWD_MAX_M,         // GFV_CBWD_MAX_M      25000  ... up to this many rows; above: the persistent fused backward
  GFV_LIM_CFWD_ON,            // GFV_CFWD            1      the column-owner small-tile forward (cfwd.hip)
  GFV_LIM_CFWD_MAX_M,         // GFV_CFWD_MAX_M      250000 (every shape: GnBlock MLPs, the encoders' narrow first layers, the decoder)
  GFV_LIM_CTRANS_ON,          // GFV_CTRANS          1      the small-tile Transolver chains (ctrans.hip)
  GFV_LIM_CTRANS_MAX_M,       // GFV_CTRANS_MAX_M    16384
  GFV_LIM_LIN1S_ON,           // GFV_LIN1S           1      the small-tile single-layer launches (lin1s.hip)
  GFV_LIM_LIN1S_MAX_M,        // GFV_LIN1S_MAX_M     16384
  GFV_LIM_COUNT
};
int gfv_internal_limit(int which);
