// mq_devbvh.h -- interface of the device-side builder of the per-frame tree (mq_devbvh.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "mq_types.h"

struct MqDevBvh {
    const MqTri* in; uint32_t n;                                          // flattened per-frame triangles (key and flags set)
    MqNode* nodes; MqLeafRec* leaves; MqTri* tris; MqShadeRec* shade;     // the scene's arrays
    uint32_t node_base, leaf_base, tri_base, node_cap;                    // where the region of this commit starts; nodes it holds
    MqSceneDev sc;                                                        // per-slot extra data and texture descriptors (shading records)
    uint32_t *keys0, *keys1, *vals0, *vals1;                              // codes and triangle numbers, before / after the sort
    int* parent; int2* child; float* box; uint32_t* flag;                 // binary tree: ids [0, n-1) internal, [n-1, 2n-1) leaves
    uint2* leaf_at;                                                       // per primitive: (leaf record, first triangle) it was given by the collapse
    uint2* queue0; uint2* queue1;                                         // wide nodes to expand: (binary id, node number within the region)
    uint32_t* ctr;                                                        // MQ_DB_* words
};
enum { MQ_DB_Q0 = 0, MQ_DB_NODES = 3, MQ_DB_LEAVES = 4, MQ_DB_ERR = 5, MQ_DB_LO = 6, MQ_DB_HI = 9, MQ_DB_MAXABS = 12, MQ_DB_DEPTH = 13, MQ_DB_PRIMS = 14 /* primitives (leaf records) */, MQ_DB_TRIS = 15, MQ_DB_WORDS = 16 };
#define MQ_DB_LEVELS 64 // collapse launches per build (a launch whose queue is empty costs ~2 us); a tree deeper than this is flagged


size_t mq_device_bvh_sort_bytes(uint32_t n);
int mq_launch_device_bvh(const MqDevBvh& A, void* sort_tmp, size_t sort_bytes, hipStream_t s);
