! TEST INFRASTRUCTURE ONLY.  bind(c) caller for the unmodified WRF-Hydro
! routine src/kernel/muskingum/MUSKINGCUNGE.f90:8-12 (submuskingcunge), so
! the deterministic original can be driven through ctypes.  Not a stand-in
! for anything: it only forwards its arguments.
subroutine c_submuskingcunge(qup, quc, qdp, ql, dt, so, dx, n, cs, bw, tw, twcc, ncc, &
                             depthp, qdc, velc, depthc) bind(c)
    use, intrinsic :: iso_c_binding, only: c_float
    use submuskingcunge_wrf_module, only: submuskingcunge
    implicit none
    real(c_float), intent(in) :: qup, quc, qdp, ql, dt, so, dx, n, cs, bw, tw, twcc, ncc, depthp
    real(c_float), intent(out) :: qdc, velc, depthc
    call submuskingcunge(qup, quc, qdp, ql, dt, so, dx, n, cs, bw, tw, twcc, ncc, &
                         depthp, qdc, velc, depthc)
end subroutine c_submuskingcunge
