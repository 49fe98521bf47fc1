#include <hip/hiprtc.h>
#include <cstdio>
#include <string>
#include <vector>
#include <fstream>
#include <sstream>
#include <chrono>
int main(int argc, char** argv) {
    std::ifstream f(argv[1]); std::stringstream ss; ss << f.rdbuf(); std::string src = ss.str();
    hiprtcProgram prog;
    hiprtcCreateProgram(&prog, src.c_str(), "jit.hip", 0, nullptr, nullptr);
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    auto t0 = std::chrono::steady_clock::now();
    hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
    auto t1 = std::chrono::steady_clock::now();
    size_t ls; hiprtcGetProgramLogSize(prog, &ls); std::string log(ls, 0); hiprtcGetProgramLog(prog, &log[0]);
    printf("result %d (%s) in %.2f s\nlog: %.2000s\n", (int)r, hiprtcGetErrorString(r), std::chrono::duration<double>(t1 - t0).count(), log.c_str());
    size_t cs = 0; hiprtcGetCodeSize(prog, &cs); printf("code size %zu\n", cs);
    return r;
}
