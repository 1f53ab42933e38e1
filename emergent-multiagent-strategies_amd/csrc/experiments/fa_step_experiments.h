// fa_config.step_kernel values of the experiment kernels (csrc/experiments/fa_step_experiments.hip).  Not in the public
// header: the product library refuses them in fa_create; only a variant library built with
//   python tools/build_variant.py experiments --add experiments/fa_step_experiments.hip
// carries the kernels (its fa_step_experiments_linked() returns 1).
#pragma once
enum { FA_KERNEL_EXP_FIRST = 64,
       FA_KERNEL_EXP_PAIRS = 64,   /* fa_step_pair_kernel (3v3): lane = (agent, partner), one wave, no workgroup barrier */
       FA_KERNEL_EXP_CHAIN = 65 }; /* fa_step_chain_kernel (3v3 / 5v5, num_steps >= 2): one workgroup barrier per step */
