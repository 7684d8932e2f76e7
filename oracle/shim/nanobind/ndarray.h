#pragma once
#include "nanobind.h"
