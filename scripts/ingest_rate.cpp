// ingest_rate.cpp -- host-inclusive rate of the batched ingest through the C ABI only.
// N channels of noise PCM arrive as one message of `chunk` samples per channel per round (the ZMQ path's shape:
// one dataReceived per channel); jaero_ingest_pump hands whole chunks to jaero_write from pinned memory.
// Prints one JSON line: samples/s including the host copy into the FIFO, the PCIe copy and the kernels, next to
// jaero_write fed from one pinned host buffer and from a device buffer.  Build (scripts/gpu_round_ingest.sh):
//   hipcc -O2 -Iinclude scripts/ingest_rate.cpp -Ljaero_amd -l:libjaero_hip.so -Wl,-rpath,$PWD/jaero_amd -o scripts/ingest_rate
#include "../include/jaero_hip.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define OK(x) do { int _r = (x); if (_r < 0) { fprintf(stderr, "%s -> %d: %s\n", #x, _r, jaero_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int nch = argc > 1 ? atoi(argv[1]) : 16384, chunk = argc > 2 ? atoi(argv[2]) : 4096, rounds = argc > 3 ? atoi(argv[3]) : 12;
    jaero_settings s{};
    s.kind = JAERO_KIND_OQPSK; s.coarsefreqest_fft_power = 14; s.freq_center = 8000; s.lockingbw = 10500; s.fb = 10500; s.Fs = 48000; s.signalthreshold = 0.65;
    jaero_ctx *bank = nullptr;
    OK(jaero_create(0, nch, &s, 0, JAERO_FLAG_EBNO, chunk, 0, &bank));
    jaero_ingest *ing = nullptr;
    OK(jaero_ingest_create(bank, chunk, 3 * chunk, &ing));
    std::vector<int16_t> msg((size_t)chunk * 64);
    unsigned x = 12345;
    for (auto &v : msg) { x = x * 1664525u + 1013904223u; v = (int16_t)((int)(x >> 18) - 8192); }
    hipStream_t st; hipStreamCreate(&st);
    auto round = [&](int r) -> int {
        for (int c = 0; c < nch; c++)
            OK(jaero_ingest_push(ing, c, msg.data() + (size_t)((c + r) & 63) * chunk, 2 * chunk, 48000));
        int n = 0;
        OK(jaero_ingest_pump(ing, 0, st, &n));
        OK(jaero_discard_softbits(bank, st));
        return 0;
    };
    for (int r = 0; r < 3; r++) if (round(r)) return 1;
    hipStreamSynchronize(st);
    double t0 = now();
    for (int r = 0; r < rounds; r++) if (round(r)) return 1;
    hipStreamSynchronize(st);
    const double t_ing = (now() - t0) / rounds;

    int16_t *pinned = nullptr, *dev = nullptr;
    hipHostMalloc((void **)&pinned, sizeof(int16_t) * (size_t)nch * chunk, hipHostMallocDefault);
    hipMalloc((void **)&dev, sizeof(int16_t) * (size_t)nch * chunk);
    for (int c = 0; c < nch; c++) for (int i = 0; i < chunk; i++) pinned[(size_t)c * chunk + i] = msg[(size_t)(c & 63) * chunk + i];
    hipMemcpy(dev, pinned, sizeof(int16_t) * (size_t)nch * chunk, hipMemcpyHostToDevice);
    double t_host = 0, t_dev = 0;
    for (int pass = 0; pass < 2; pass++)
    {
        for (int r = 0; r < 3; r++) { OK(jaero_write(bank, pass ? dev : pinned, chunk, JAERO_PCM_CHANNEL_MAJOR, pass, st)); OK(jaero_discard_softbits(bank, st)); }
        hipStreamSynchronize(st);
        t0 = now();
        for (int r = 0; r < rounds; r++) { OK(jaero_write(bank, pass ? dev : pinned, chunk, JAERO_PCM_CHANNEL_MAJOR, pass, st)); OK(jaero_discard_softbits(bank, st)); }
        hipStreamSynchronize(st);
        (pass ? t_dev : t_host) = (now() - t0) / rounds;
    }
    const double S = (double)nch * chunk;
    printf("{\"channels\": %d, \"chunk\": %d, \"rounds\": %d, \"ingest_Msamples_s\": %.1f, \"pinned_write_Msamples_s\": %.1f, \"device_write_Msamples_s\": %.1f, "
           "\"ms_per_round\": {\"ingest\": %.2f, \"pinned_write\": %.2f, \"device_write\": %.2f}, \"host_threads\": 1}\n",
           nch, chunk, rounds, S / t_ing / 1e6, S / t_host / 1e6, S / t_dev / 1e6, t_ing * 1e3, t_host * 1e3, t_dev * 1e3);
    jaero_ingest_destroy(ing);
    jaero_destroy(bank);
    return 0;
}
