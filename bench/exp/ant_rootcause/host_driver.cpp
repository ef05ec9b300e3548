// csrc/ant_sim.h stepped on the host from the files of dump_params.py: the build that the sanitizers see.
//   amdclang++ -O3 -g -fsanitize=address,undefined -fno-sanitize-recover=all -Wall -Wextra -I../../../d4pg_pytorch_amd/csrc host_driver.cpp
//   amdclang++ -O3 -g -fsanitize=memory -DASIM_NO_SINCOS ...   (sincosf's outputs are written by uninstrumented libm)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ant_sim.h"

template <typename T>
static std::vector<T> slurp(const char* dir, const char* name) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<T> v(n / sizeof(T));
    if (fread(v.data(), 1, n, f) != (size_t)n) exit(2);
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    using namespace ant;
    const char* dir = argc > 1 ? argv[1] : ".";
    auto pb = slurp<char>(dir, "params.bin");
    if (pb.size() != sizeof(AntParams)) { fprintf(stderr, "params.bin: %zu bytes, AntParams %zu\n", pb.size(), sizeof(AntParams)); return 2; }
    const AntParams* P = (const AntParams*)pb.data();
    auto st = slurp<float>(dir, "states.bin");
    auto act = slurp<float>(dir, "actions.bin");
    const int E = (int)st.size() / STRIDE, T = (int)act.size() / (E * 8);
    double sum = 0.0;
    int terms = 0;
    for (int t = 0; t < T; ++t)
        for (int e = 0; e < E; ++e) {
            float* s = st.data() + (size_t)e * STRIDE;
            float pos[3], quat[4], qj[8], u[N], wrench[L][6], rew, obs[S];
            bool term;
            for (int i = 0; i < 3; ++i) pos[i] = s[i];
            for (int i = 0; i < 4; ++i) quat[i] = s[3 + i];
            for (int i = 0; i < 8; ++i) qj[i] = s[7 + i];
            for (int i = 0; i < N; ++i) u[i] = s[15 + i];
            for (int i = 0; i < L; ++i)
                for (int k = 0; k < 6; ++k) wrench[i][k] = 0.f;
            advance(P, pos, quat, qj, u, act.data() + ((size_t)t * E + e) * 8, wrench, rew, term);
            observe(pos, quat, qj, u, wrench, obs);
            for (int i = 0; i < 3; ++i) s[i] = pos[i];
            for (int i = 0; i < 4; ++i) s[3 + i] = quat[i];
            for (int i = 0; i < 8; ++i) s[7 + i] = qj[i];
            for (int i = 0; i < N; ++i) s[15 + i] = u[i];
            for (int i = 0; i < S; ++i) sum += obs[i];
            sum += rew;
            terms += term;
        }
    printf("E=%d T=%d checksum=%.9g terminations=%d\n", E, T, sum, terms);
    return 0;
}
