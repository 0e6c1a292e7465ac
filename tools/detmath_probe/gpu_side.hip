#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include "../../include/ig_detmath.h"
extern "C" void cpu_eval(const float* s, const float* stot, const int* ob, size_t n, ig_params p, const double* lgf,
              float* o_pow, float* o_exp, double* o_log10, float* o_r, float* o_rc, double* o_term, long long* o_q);
__global__ void gpu_eval(const float* s, const float* stot, const int* ob, size_t n, ig_params p, const double* lgf,
              float* o_pow, float* o_exp, double* o_log10, float* o_r, float* o_rc, double* o_term, long long* o_q)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    o_pow[i] = ig_powf(s[i], p.slope, ig_tab());
    o_exp[i] = ig_expf(-s[i] * 0.01f, ig_tab());
    o_log10[i] = ig_log10((double)s[i], ig_tab());
    o_r[i] = ig_rippe(s[i], p, ig_tab());
    o_rc[i] = ig_rippe_circ(s[i], stot[i], p, ig_tab());
    double lg = ig_lgfact(ob[i] > 0 ? ob[i] : 1, lgf, ig_tab());
    o_term[i] = ig_pixel_term(o_r[i], o_rc[i], ob[i], lg, ig_tab());
    o_q[i] = ig_quantize(o_term[i]);
}
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2);} }while(0)
template<class T> T* up(const std::vector<T>& v){T* d; CK(hipMalloc(&d, v.size()*sizeof(T))); CK(hipMemcpy(d, v.data(), v.size()*sizeof(T), hipMemcpyHostToDevice)); return d;}
template<class T> size_t cmp(const char* name, const std::vector<T>& a, T* d){ std::vector<T> b(a.size()); CK(hipMemcpy(b.data(), d, a.size()*sizeof(T), hipMemcpyDeviceToHost)); size_t bad=0; for(size_t i=0;i<a.size();i++) if(memcmp(&a[i],&b[i],sizeof(T))) {bad++; } printf("%-8s mismatches %zu / %zu\n", name, bad, a.size()); return bad;}
int main(){
    size_t n = 1<<22;
    std::vector<float> s(n), st(n); std::vector<int> ob(n);
    unsigned long long x = 88172645463325252ULL;
    auto rnd=[&](){ x ^= x<<13; x ^= x>>7; x ^= x<<17; return x; };
    for(size_t i=0;i<n;i++){ double u=(rnd()>>11)*(1.0/9007199254740992.0); double v=(rnd()>>11)*(1.0/9007199254740992.0);
        s[i]=(float)std::exp(std::log(1e-3)+u*std::log(1e7)); if(i%1000==0) s[i]=0.f; if(i%1001==0) s[i]=-1.f; if (i%1003==0) { unsigned r=(unsigned)rnd(); memcpy(&s[i], &r, 4);} 
        st[i]=(float)(s[i]*(0.5+2*v)); ob[i]=(int)(rnd()%40); if(i%17==0) ob[i]=(int)(rnd()%5000);}
    ig_params p={50.f,9.6f,0.f,-1.5f,2.f,1500.f,3.0e5f,5e-3f}; p.c1=(float)(0.53*std::pow(9.6/50.,-1.5)*std::pow(50.,-3));
    std::vector<double> lgf(15); for(int k=0;k<15;k++){ double f=1; for(int c=1;c<=k;c++) f*=c; lgf[k]=std::log10(f);} 
    std::vector<float> a_pow(n),a_exp(n),a_r(n),a_rc(n); std::vector<double> a_l(n),a_t(n); std::vector<long long> a_q(n);
    cpu_eval(s.data(),st.data(),ob.data(),n,p,lgf.data(),a_pow.data(),a_exp.data(),a_l.data(),a_r.data(),a_rc.data(),a_t.data(),a_q.data());
    float *ds=up(s),*dst=up(st); int* dob=up(ob); double* dl=up(lgf);
    float *g_pow,*g_exp,*g_r,*g_rc; double *g_l,*g_t; long long* g_q;
    CK(hipMalloc(&g_pow,n*4));CK(hipMalloc(&g_exp,n*4));CK(hipMalloc(&g_r,n*4));CK(hipMalloc(&g_rc,n*4));CK(hipMalloc(&g_l,n*8));CK(hipMalloc(&g_t,n*8));CK(hipMalloc(&g_q,n*8));
    hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for(int it=0;it<3;it++){ hipEventRecord(e0); hipLaunchKernelGGL(gpu_eval, dim3((n+255)/256), dim3(256), 0, 0, ds,dst,dob,n,p,dl,g_pow,g_exp,g_l,g_r,g_rc,g_t,g_q); hipEventRecord(e1); CK(hipDeviceSynchronize()); float ms; hipEventElapsedTime(&ms,e0,e1); printf("gpu_eval %.3f ms (%.1f Mevals/s)\n", ms, n/ms/1e3);} 
    size_t bad=0; bad+=cmp("powf",a_pow,g_pow); bad+=cmp("expf",a_exp,g_exp); bad+=cmp("log10",a_l,g_l); bad+=cmp("rippe",a_r,g_r); bad+=cmp("rippe_c",a_rc,g_rc); bad+=cmp("term",a_t,g_t); bad+=cmp("quant",a_q,g_q);
    /* accuracy vs libm on the CPU side (informational) */
    double maxrel=0; size_t cnt=0; for(size_t i=0;i<n;i++){ if(!(s[i]>0)||!std::isfinite(s[i])) continue; double ref=std::pow((double)s[i],-1.5); if(!(ref>1e-37&&ref<1e37)) continue; double rel=std::fabs(a_pow[i]-ref)/ref; if(rel>maxrel) maxrel=rel; float fr=(float)ref; if(fr!=a_pow[i]) cnt++; }
    printf("powf max rel err vs double pow %.3e ; not-correctly-rounded count %zu\n", maxrel, cnt);
    double maxabs=0; for(size_t i=0;i<n;i++){ if(!(s[i]>0)||!std::isfinite(s[i])) continue; double d=std::fabs(a_l[i]-std::log10((double)s[i])); if(d>maxabs) maxabs=d;} printf("log10 max abs err vs libm %.3e\n", maxabs);
    printf(bad? "RESULT: MISMATCH\n":"RESULT: BIT-EXACT\n"); return bad?1:0; }
