#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
static inline double rcp_approx(double u){ float f = 1.0f/(float)u; return (double)f; } /* ~24-bit seed like a hw rcp */
static inline double rcp_nr(double u){ double y = rcp_approx(u); double e = fma(-u,y,1.0); y = fma(y,e,y); e = fma(-u,y,1.0); y = fma(y,e,y); e = fma(-u,y,1.0); y = fma(y,e,y); return y; }
static double my_exp_neg(double a){ /* exp(-a), a>=0 */
  if (a > 800.0) a = 800.0;
  const double L2E = 1.4426950408889634074, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  double kf = rint(a*L2E);
  double r = fma(kf, LN2_HI, -a);       /* k*ln2_hi - a  = -(a - k ln2_hi) */
  r = fma(kf, LN2_LO, r);               /* r = k*ln2 - a, |r| <= ln2/2 */
  /* exp(r) Taylor degree 13 */
  static const double c[] = {1.0/6227020800.0,1.0/479001600.0,1.0/39916800.0,1.0/3628800.0,1.0/362880.0,1.0/40320.0,1.0/5040.0,1.0/720.0,1.0/120.0,1.0/24.0,1.0/6.0,0.5};
  double p = c[0];
  for (int i=1;i<12;i++) p = fma(p,r,c[i]);
  p = fma(p, r*r, r);  /* r + r^2 * P */
  p = p + 1.0;
  return ldexp(p, -(int)kf);
}
static double my_log_1_2(double u){ /* log(u), u in [1,2] */
  const double LN2 = 0.693147180559945309417;
  int k = u > 1.4142135623730951;
  double m = k ? 0.5*u : u;
  double num = m - 1.0, den = m + 1.0;
  double y = rcp_nr(den); double s = num * y; double slo = fma(-s, den, num) * y;
  /* one correction step for the quotient: s += (num - s*den)*rcp */
  double s2 = s*s;
  static const double q[] = {2.0/21.0,2.0/19.0,2.0/17.0,2.0/15.0,2.0/13.0,2.0/11.0,2.0/9.0,2.0/7.0,2.0/5.0,2.0/3.0};
  double p = q[0];
  for (int i=1;i<10;i++) p = fma(p,s2,q[i]);
  const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  double lo = fma(s*s2, p, (k ? LN2_LO : 0.0) + 2.0*slo);
  double res = 2.0*s + lo;
  return k ? res + LN2_HI : res;
}
static double ulp_err(double got, long double want){ if (want==0) return got==0?0:1e9; double u = nextafter((double)want, INFINITY) - (double)want; return (double)fabsl(((long double)got - want)/u); }
int main(){
  double maxe_exp=0,maxe_log=0,maxe_ce=0,maxe_sig=0; srand(1);
  for (long i=0;i<4000000;i++){
    double a = (i%4==0)? (rand()/(double)RAND_MAX)*50.0 : (i%4==1)? (rand()/(double)RAND_MAX)*2.0 : (i%4==2)? exp(-(rand()/(double)RAND_MAX)*40.0) : (rand()/(double)RAND_MAX)*750.0;
    double e = my_exp_neg(a);
    long double el = expl(-(long double)a);
    double ee = ulp_err(e, el); if (ee>maxe_exp) maxe_exp=ee;
    double u = 1.0 + e;
    double lg = my_log_1_2(u);
    long double lgl = logl((long double)u);
    double le = ulp_err(lg, lgl); if (le>maxe_log) maxe_log=le;
    /* composite vs reference formula evaluated in long double on the same rounding of (1+e_ref) */
    double e_ref = exp(-a); double ce_ref = log(1.0 + e_ref);
    double ce_e = ulp_err(lg, (long double)ce_ref); if (ce_e>maxe_ce) maxe_ce=ce_e;
    double sg = rcp_nr(u); double se = ulp_err(sg, 1.0L/(long double)u); if (se>maxe_sig) maxe_sig=se;
  }
  printf("max ulp: exp %.2f log(u) %.2f ce-vs-glibc %.2f rcp %.2f\n", maxe_exp, maxe_log, maxe_ce, maxe_sig);
  return 0;
}
