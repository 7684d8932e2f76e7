// container shim header: see ../../xt_shim.hpp (stand-in for xtensor/generators/xrandom.hpp)
#pragma once
#include "../../xt_shim.hpp"
