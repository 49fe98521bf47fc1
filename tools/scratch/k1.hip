extern "C" __global__ void k(uint64_t* p, unsigned long long* q) { p[threadIdx.x] += 1; q[blockIdx.x] = __brevll(q[0]); }
