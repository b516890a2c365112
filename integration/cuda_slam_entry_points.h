// The three entry points of the reference's GPU project, with the signatures of source/cuda-slam/icpcuda.cuh:5-11,
// cpdcuda.cuh:5-17 and nicpcuda.cuh:5-15.  In the reference tree those headers declare them (and pull in CUDA/thrust, which this
// image does not have); the adapter check in oracle/Makefile uses this header in their place -- the reference's OWN types
// (Common::Point_f, glm, Common::ApproximationType from source/common) are used as they are.
#pragma once
#include <utility>
#include <vector>

#include "common.h"

std::pair<glm::mat3, glm::vec3> GetCudaIcpTransformationMatrix(const std::vector<Common::Point_f>& cloudBefore,
                                                               const std::vector<Common::Point_f>& cloudAfter, float eps,
                                                               int maxIterations, int* iterations, float* error);

std::pair<glm::mat3, glm::vec3> GetCudaCpdTransformationMatrix(const std::vector<Common::Point_f>& cloudBefore,
                                                               const std::vector<Common::Point_f>& cloudAfter, float eps, float weight,
                                                               bool const_scale, int maxIterations, float tolerance,
                                                               Common::ApproximationType fgt, int* iterations, float* error,
                                                               const float& ratioOfFarField, const float& orderOfTruncation);

std::pair<glm::mat3, glm::vec3> GetCudaNicpTransformationMatrix(const std::vector<Common::Point_f>& before,
                                                                const std::vector<Common::Point_f>& after, float eps, int maxRepetitions,
                                                                int batchSize, Common::ApproximationType approximationType,
                                                                const int subcloudSize, int* repetitions, float* error);
