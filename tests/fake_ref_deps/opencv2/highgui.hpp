#pragma once
#include <opencv2/core.hpp>
