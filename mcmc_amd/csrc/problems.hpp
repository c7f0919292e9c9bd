// What the step-wise Metropolis-Hastings engine (gmh_api.hip) needs to know about the problem
// kinds that live in other translation units.
#pragma once
#include "beam_device.hpp"
#include "common.hpp"

namespace mi355 {

// binnedCl: number of parameters and the spectrum type of each (device array), cl_api.hip
int cl_problem_info(const mi355_ctx* c, int* npar, const int32_t** cltype_dev);
// BeamFit: number of maps and the prior box, beam_api.hip
int beam_problem_info(const mi355_ctx* c, int64_t* M, BeamPriorBox* box);

}  // namespace mi355
