cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for prio in 0 3; do for gate in 0 64 88 104; do
echo "== prio $prio gate $gate"; DANET_LSTM_PRIO=$prio DANET_FOLLOW_GATE=$gate timeout 300 python tools/follow_probe.py 2>&1 | grep "^dX\|^dW\|^BPTT"
done; done
echo "== neighbour probe prio 0"; DANET_LSTM_PRIO=0 timeout 300 python tools/bptt_neighbour_probe.py 2>&1 | tail -7
echo "== neighbour probe prio 3"; DANET_LSTM_PRIO=3 timeout 300 python tools/bptt_neighbour_probe.py 2>&1 | tail -7
