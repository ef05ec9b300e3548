// The host build of the call in ant_step_bisect.hip.
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
    if (argc < 3) return 2;
    auto pb = slurp<char>(argv[1], "params.bin");
    auto st = slurp<float>(argv[1], "states_contact.bin");
    auto act = slurp<float>(argv[1], "actions.bin");
    const AntParams* P = (const AntParams*)pb.data();
    const int E = (int)st.size() / STRIDE;
    std::vector<float> out((size_t)E * 30);
    for (int e = 0; e < E; ++e) {
        const float* s = st.data() + (size_t)e * STRIDE;
        float pos[3], quat[4], qj[8], u[N], wrench[L][6], rew;
        bool term;
        for (int i = 0; i < 3; ++i) pos[i] = s[i];
        for (int i = 0; i < 4; ++i) quat[i] = s[3 + i];
        for (int i = 0; i < 8; ++i) qj[i] = s[7 + i];
        for (int i = 0; i < N; ++i) u[i] = s[15 + i];
        for (int i = 0; i < L; ++i)
            for (int k = 0; k < 6; ++k) wrench[i][k] = 0.f;
        advance(P, pos, quat, qj, u, act.data() + (size_t)e * 8, wrench, rew, term);
        float* o = out.data() + (size_t)e * 30;
        for (int i = 0; i < 3; ++i) o[i] = pos[i];
        for (int i = 0; i < 4; ++i) o[3 + i] = quat[i];
        for (int i = 0; i < 8; ++i) o[7 + i] = qj[i];
        for (int i = 0; i < N; ++i) o[15 + i] = u[i];
        o[29] = rew;
    }
    FILE* f = fopen(argv[2], "wb");
    fwrite(out.data(), 4, out.size(), f);
    fclose(f);
    return 0;
}
