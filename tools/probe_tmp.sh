P="timeout -k 10 150 python tools/stackw_probe.py"
$P fwd_t
QPN_STACK_WAVE_FWD=0 $P fwd_old
QPN_STACK_WGS=384 $P fwd_t_384
QPN_STACK_WGS=768 $P fwd_t_768
