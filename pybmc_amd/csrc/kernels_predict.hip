// Posterior-predictive kernels (reference pybmc/sampling_utils.py:40-84) -- placeholder
// translation unit, filled in once the Gibbs loop is parity-green.
#include "bmc_dev.h"
#include "bmc_launch.h"
namespace bmc {}
