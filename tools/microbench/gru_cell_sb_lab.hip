// Phase stamps of the split-bf16 rollout GRU cell (k_gru_cell_sb): the product source compiled with -DSBC_STAMP, one paired launch
// at the benchmark's 32 768 rows; prints per wave of one workgroup the cycles spent in the products, staging, barrier wait and
// gate math of the tile loop, and the shader clock.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSBC_STAMP -I include tools/microbench/gru_cell_sb_lab.hip -o tools/microbench/gru_cell_sb_lab
#include "../../distributed_multi_agent_reinforcement_learning_amd/csrc/mappo_ops.hip"

#include <cstdio>
#include <vector>

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32768;
    float *buf[2][8];
    std::vector<float> h((size_t)B * 128);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    mo_gru_cell_net nets[2];
    for (int k = 0; k < 2; k++) {
        const size_t sz[7] = {(size_t)B * 128, (size_t)B * 128, 384 * 128, 384 * 128, 384, 384, (size_t)B * 128};
        for (int j = 0; j < 7; j++) {
            (void)hipMalloc(&buf[k][j], sz[j] * 4);
            (void)hipMemcpy(buf[k][j], h.data(), sz[j] * 4, hipMemcpyHostToDevice);
        }
        nets[k] = mo_gru_cell_net{buf[k][0], buf[k][1], buf[k][2], buf[k][3], buf[k][4], buf[k][5], buf[k][6]};
    }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) gru_cell_split_fwd_multi(2, nets, B, 128, 0);
    (void)hipEventRecord(e0, 0);
    for (int w = 0; w < 20; w++) gru_cell_split_fwd_multi(2, nets, B, 128, 0);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%d rows, 2 cells: %.1f us per launch (stamped build)\n", B, ms / 20 * 1e3);
    unsigned long long st[8][8];
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(sbc_stamps), sizeof st);
    for (int w = 0; w < 8; w++) {
        const double n = (double)st[w][4];
        printf("wave %d (group %d role %d): per tile products %.0f  staging %.0f  barrier wait %.0f  gate math %.0f cycles | %llu tiles, loop %.1f us at %.0f MHz\n", w, w & 3,
               w >> 2, st[w][0] / n, st[w][1] / n, st[w][2] / n, st[w][3] / n, st[w][4], st[w][6] / 100.0, st[w][5] / (st[w][6] / 100.0));
    }
    return 0;
}
