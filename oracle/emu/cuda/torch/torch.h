// just enough of the names common.h mentions in (unexpanded) macros and one using-declaration
#pragma once
#include "../cuda_runtime.h"
namespace at { enum class ScalarType { Bool, Int, Float, Double }; }
namespace torch { struct Tensor {}; }
#define TORCH_CHECK(...)
