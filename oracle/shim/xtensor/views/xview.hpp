// container shim header: see ../../xt_shim.hpp (stand-in for xtensor/views/xview.hpp)
#pragma once
#include "../../xt_shim.hpp"
