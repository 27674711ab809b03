// double instantiation of the step-loop kernels: the parity mode.  Built with -ffp-contract=off so
// that a*b+c rounds twice like the NumPy oracle (and nengo's CPU backend) does.
#include "ssn_kernels.hpp"
namespace ssn { SSN_INSTANTIATE(double) }
