// dev-only: tau0[i] = k-th smallest D(i, j) over j within +-W positions of i in the order perm (variant A and B)
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <omp.h>
static int cmpd(const void*a,const void*b){double x=*(double*)a,y=*(double*)b;return x<y?-1:x>y;}
int main(int argc,char**argv){
  int64_t n=atol(argv[1]); int K=3; int W=atoi(argv[4]); int kk=atoi(argv[5]); int variant=atoi(argv[6]); // 0=A 1=B
  FILE*f=fopen(argv[2],"rb"); int B; fread(&B,4,1,f); int*sup=malloc(B*4); fread(sup,4,B,f);
  uint16_t*H=malloc((size_t)n*K*B*2); fread(H,2,(size_t)n*K*B,f); fclose(f);
  int32_t*perm=malloc(n*4); f=fopen(argv[3],"rb"); fread(perm,4,n,f); fclose(f);
  int64_t*rp=malloc((n+3)*8); f=fopen("ba100k.bin","rb"); fread(rp,8,n+3,f); fclose(f);
  double*deg=malloc(n*8); for(int64_t i=0;i<n;i++)deg[i]=(double)(rp[i+3]-rp[i+2]);
  int L=variant?2:3; int C=variant?B:(B-1);
  double*X=malloc((size_t)n*L*C*8); double*nk=malloc((size_t)n*L*8);
  #pragma omp parallel for
  for(int64_t i=0;i<n;i++) for(int k=0;k<L;k++){ uint16_t*h=H+((size_t)i*K+k)*B; double tot=0; for(int b=0;b<B;b++)tot+=h[b]; nk[i*L+k]=tot;
     double*x=X+((size_t)i*L+k)*C; 
     if(!variant){ double c=0; for(int b=0;b<B-1;b++){c+=h[b]; x[b]=c/tot*(sup[b+1]-sup[b]);} }
     else { x[0]=tot*sup[0]; double t=tot; for(int b=0;b<B-1;b++){t-=h[b]; x[1+b]=t*(sup[b+1]-sup[b]);} } }
  double*tau=malloc(n*8);
  #pragma omp parallel
  { double*d=malloc((2*W+1)*8);
    #pragma omp for schedule(dynamic,64)
    for(int64_t p=0;p<n;p++){ int64_t i=perm[p]; int m=0;
      for(int64_t q=p-W;q<=p+W;q++){ if(q<0||q>=n||q==p)continue; int64_t j=perm[q]; double D=fabs(deg[i]-deg[j]);
        for(int k=0;k<L;k++){ double*xi=X+((size_t)i*L+k)*C,*xj=X+((size_t)j*L+k)*C; double s=0; for(int c=0;c<C;c++)s+=fabs(xi[c]-xj[c]);
          if(variant){double mx=nk[i*L+k]>nk[j*L+k]?nk[i*L+k]:nk[j*L+k]; s/=mx;} D+=s; }
        d[m++]=D; }
      qsort(d,m,8,cmpd); tau[i]=d[kk-1<m?kk-1:m-1]; }
  }
  f=fopen(argv[7],"wb"); fwrite(tau,8,n,f); fclose(f); return 0; }
