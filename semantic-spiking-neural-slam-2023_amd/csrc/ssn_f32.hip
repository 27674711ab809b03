// float instantiation of the step-loop kernels: the fast mode (FMA contraction allowed).
#include "ssn_kernels.hpp"
namespace ssn { SSN_INSTANTIATE(float) }

#ifdef SSN_PROGRAM_STAMPS
extern "C" int ssn_debug_program_stamps(unsigned long long* out, int n) { return (int)ssn::read_program_stamps(out, n); }
#endif
#ifdef SSN_ROUND_STAMPS
extern "C" long long ssn_debug_round_stamps(unsigned long long* out, long long cap, int reset) { return ssn::read_round_stamps(out, cap, reset); }
#endif
#ifdef SSN_BLOCK_STAMPS
extern "C" int ssn_debug_block_stamps(unsigned long long* out, int reset) { return (int)ssn::read_block_stamps(out, reset); }
#endif
