// Development probe (GPU box): host-side cost of hipSetDevice and of filling / reading a 4.6 KB struct in hipHostMalloc memory
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
struct Item { char b[4584]; };
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipSetDevice(0);
    const int N = 128, R = 200;
    Item *pin = nullptr, *pin_wc = nullptr;
    hipHostMalloc((void **)&pin, sizeof(Item) * N, hipHostMallocDefault);
    hipHostMalloc((void **)&pin_wc, sizeof(Item) * N, hipHostMallocWriteCombined);
    Item *heap = (Item *)malloc(sizeof(Item) * N);
    Item src; memset(&src, 3, sizeof(src));
    volatile long sink = 0;
    for (auto *p : {heap, pin, pin_wc}) {
        double t0 = now();
        for (int r = 0; r < R; r++) for (int i = 0; i < N; i++) { p[i] = Item{}; memcpy(&p[i], &src, sizeof(Item)); }
        double t1 = now();
        long s = 0;
        for (int r = 0; r < R; r++) for (int i = 0; i < N; i++) for (int k = 0; k < 4584; k += 64) s += p[i].b[k];
        double t2 = now();
        sink += s;
        printf("%s: fill %.3f us/item, strided read %.3f us/item\n", p == heap ? "heap" : p == pin ? "pinned default" : "pinned write-combined", (t1 - t0) / (R * N), (t2 - t1) / (R * N));
    }
    double t0 = now();
    for (int i = 0; i < 100000; i++) hipSetDevice(0);
    double t1 = now();
    printf("hipSetDevice %.3f us\n", (t1 - t0) / 100000);
    hipStream_t s; hipStreamCreate(&s);
    t0 = now();
    for (int i = 0; i < 100000; i++) (void)hipGetLastError();
    t1 = now();
    printf("hipGetLastError %.3f us\n", (t1 - t0) / 100000);
    t0 = now();
    for (int i = 0; i < 20000; i++) (void)hipStreamQuery(s);
    t1 = now();
    printf("hipStreamQuery(idle) %.3f us\n", (t1 - t0) / 20000);
    return 0;
}
