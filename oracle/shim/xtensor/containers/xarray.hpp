// container shim header: see ../../xt_shim.hpp (stand-in for xtensor/containers/xarray.hpp)
#pragma once
#include "../../xt_shim.hpp"
