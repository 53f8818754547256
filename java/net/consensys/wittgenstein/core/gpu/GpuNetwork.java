package net.consensys.wittgenstein.core.gpu;

import java.lang.reflect.Field;
import java.util.ArrayList;
import java.util.Iterator;
import java.util.List;
import java.util.Random;
import java.util.concurrent.atomic.AtomicLong;
import net.consensys.wittgenstein.core.GpuNodeAccess;
import net.consensys.wittgenstein.core.Network;
import net.consensys.wittgenstein.core.Node;
import net.consensys.wittgenstein.core.messages.ConditionalTask;
import net.consensys.wittgenstein.core.messages.Message;
import net.consensys.wittgenstein.core.messages.PeriodicTask;
import net.consensys.wittgenstein.core.messages.Task;

/**
 * core.Network over libwittgpu.so. Protocols keep calling the methods they call today (C/Network.java:341-447 send /
 * sendAll / sendArriveAt, :505-531 registerTask / registerPeriodicTask / registerConditionalTask, :304-338 run / runH /
 * runMs); what moves to the MI355X is the message queue with its LIFO / multi-destination ordering (MessageStorage
 * :116-299, Envelope.java:57-301), NetworkLatency sampling (:449-487) and the shared rd's draws for send().
 *
 * <p>Two modes. HOST-CALLBACK (the default, {@link #GpuNetwork(String)}): any Protocol written against Network runs
 * unchanged — Message.action() stays in Java, the engine hands back deliveries in the reference's total order, a
 * millisecond per native call (wg_step_begin / wg_step_end), and takes the action()s' pushes back with the seeds this
 * class drew from {@code rd} in the reference's order. RESIDENT ({@link #attachResident(long)}): the protocol's
 * action() code runs on the device (Handel, GSFSignature, PingPong, San Fermin, Casper IMD, P2PFlood — see
 * WittGpu.load*); runMs is then one native call and the node counters / protocol fields are read back on demand.
 *
 * <p>The Python class wittgenstein_amd/hostnet.py is this class line for line and is what the repository's tests run
 * against the oracle (tests/test_gpu_hostmode.py); no JDK exists in its build image, so this file is not compiled there.
 */
public class GpuNetwork<TN extends Node> extends Network<TN> {
  /** the engine (wg_engine*) */
  public long handle;

  private boolean resident = false, ready = false;
  private final String latencyName;
  // handle -> Message, for as long as an envelope of it is in flight: the engine reports the end of every envelope
  // (WittGpu.hostReleased) and the object is let go with its last one — where the reference lets go of its Envelope. (Round 4
  // kept every Message of a run: 1e7 - 1e8 of them in a Handel run of a few thousand nodes, and an int that overflows.)
  private final ArrayList<Message<? extends TN>> byHandle = new ArrayList<>(); // (0 unused)
  private final java.util.IdentityHashMap<Message<? extends TN>, Integer> handleOfObj = new java.util.IdentityHashMap<>();
  private int[] refs = new int[1024]; // handle -> envelopes in flight (a re-armed PeriodicTask, a Message sent twice: one handle)
  private int[] free = new int[1024];
  private int nFree = 0;
  private final int[] released = new int[1024];
  private final List<ConditionalTask<TN>> condTasks = new ArrayList<>(); // Network.conditionalTasks is private there

  // a batched step (wg_step_begin .. wg_step_end): the pushes of the action()s, ten ints an op (wg_step_op)
  private int[] ops = null;
  private int nops = 0, cur = 0;
  private int[] opDests = new int[64];
  private int nOpDests = 0;
  private final int[] batch;

  public GpuNetwork(String networkLatencyName) {
    this(networkLatencyName, null, null, 4096);
  }

  /** cfgInts / cfgLongs: wg_config (WittGpu.create); batchCap: deliveries handed out per native call */
  public GpuNetwork(String networkLatencyName, int[] cfgInts, long[] cfgLongs, int batchCap) {
    this.latencyName = networkLatencyName;
    this.handle = WittGpu.create(cfgInts, cfgLongs, null);
    this.batch = new int[6 * batchCap];
    this.byHandle.add(null);
  }

  /** the protocol's resident form was loaded into {@code h} (WittGpu.load* / WittGpu.host*Create): this Network drives it */
  public GpuNetwork<TN> attachResident(long h) {
    if (handle != 0 && handle != h) WittGpu.destroy(handle);
    handle = h;
    resident = ready = true;
    return this;
  }

  public void close() {
    if (handle != 0) WittGpu.destroy(handle);
    handle = 0;
  }

  // ---- rd: Network.rd is `public final Random rd = new Random(0)` (C/Network.java:32). Its 48-bit state is copied to the
  // engine before a native call that draws (wg_send, wg_run_ms of a resident protocol) and read back after it, so that the
  // one stream serves Java code and engine alike. (java.util.Random keeps it in the private AtomicLong `seed`, scrambled
  // only by setSeed: JDK 9 — the reference's target — allows the reflective access; later JDKs need
  // --add-opens java.base/java.util=ALL-UNNAMED.)
  private static final Field SEED;

  static {
    try {
      SEED = Random.class.getDeclaredField("seed");
      SEED.setAccessible(true);
    } catch (ReflectiveOperationException e) {
      throw new ExceptionInInitializerError(e);
    }
  }

  static long stateOf(Random r) {
    try {
      return ((AtomicLong) SEED.get(r)).get();
    } catch (IllegalAccessException e) {
      throw new IllegalStateException(e);
    }
  }

  static void setState(Random r, long s48) {
    try {
      ((AtomicLong) SEED.get(r)).set(s48);
    } catch (IllegalAccessException e) {
      throw new IllegalStateException(e);
    }
  }

  private void rdToEngine() {
    WittGpu.rngSetState(handle, stateOf(rd));
  }

  private void rdFromEngine() {
    setState(rd, WittGpu.rngGetState(handle));
  }

  // ---- nodes: the Java objects stay; the engine gets their coordinates once, before the first send / run
  private void start() {
    if (ready) return;
    int n = allNodes.size();
    int[] x = new int[n], y = new int[n], extra = new int[n];
    byte[] down = new byte[n], byz = new byte[n];
    double[] speed = new double[n];
    for (int i = 0; i < n; i++) {
      TN nd = allNodes.get(i);
      if (nd == null || nd.nodeId != i) throw new IllegalStateException("node ids must be dense from 0 (C/Network.java:25-29)");
      x[i] = nd.x;
      y[i] = nd.y;
      extra[i] = nd.extraLatency;
      down[i] = (byte) (nd.isDown() ? 1 : 0);
      byz[i] = (byte) (nd.byzantine ? 1 : 0);
      speed[i] = nd.speedRatio;
    }
    WittGpu.setLatencyByName(handle, latencyName);
    WittGpu.addNodes(handle, x, y, extra, down, byz, speed);
    WittGpu.loadHost(handle);
    ready = true;
  }

  /** the handle an envelope of {@code m} travels under: one more envelope of it in flight */
  private int handleOf(Message<? extends TN> m) {
    Integer known = handleOfObj.get(m);
    int h;
    if (known != null) {
      h = known;
    } else {
      if (nFree > 0) {
        h = free[--nFree];
        byHandle.set(h, m);
      } else {
        byHandle.add(m);
        h = byHandle.size() - 1;
        if (h >= refs.length) refs = java.util.Arrays.copyOf(refs, refs.length * 2);
      }
      handleOfObj.put(m, h);
      refs[h] = 0;
    }
    refs[h]++;
    return h;
  }

  /** A native call that creates ONE envelope of `m`: the reference handleOf took for it is given back if the engine refuses the
   *  call (IllegalArgumentException / IllegalStateException / OutOfMemoryError mapped from WG_EINVAL / WG_ESTATE / WG_ENOMEM) — no
   *  envelope exists then, wg_host_released will never report it, and the Message would stay pinned and its handle lost for the
   *  rest of the run (ADVICE.md round 5; hostnet.HostNetwork._native is the same rule). */
  private interface NativeCall { void with(int handleOfMessage); }

  private void underHandle(Message<? extends TN> m, NativeCall call) {
    final int h = handleOf(m);
    try {
      call.with(h);
    } catch (RuntimeException | Error e) {
      unref(h);
      throw e;
    }
  }

  private void unref(int h) {
    if (--refs[h] > 0) return;
    handleOfObj.remove(byHandle.get(h));
    byHandle.set(h, null);
    if (nFree == free.length) free = java.util.Arrays.copyOf(free, free.length * 2);
    free[nFree++] = h;
  }

  /** envelopes that ended since the last call (wg_host_released): their Messages are forgotten with their last envelope */
  private void drainReleased() {
    int n;
    do {
      n = WittGpu.hostReleased(handle, released);
      for (int i = 0; i < n; i++) unref(released[i]);
    } while (n == released.length);
  }

  /** Node.stop() / start() of a node of a running network (C/Node.java:120-131) */
  public void setDown(TN node, boolean down) {
    if (down) node.stop();
    else node.start();
    if (ready) WittGpu.setNodeDown(handle, node.nodeId, down);
  }

  // ---- sends (C/Network.java:341-447). The 3-argument overloads of the superclass funnel into these.
  private void checkFrom(TN fromNode) {
    if (fromNode.nodeId >= allNodes.size() || getNodeById(fromNode.nodeId) != fromNode)
      throw new IllegalArgumentException("The from node is not in the network. From=" + fromNode);
  }

  private void op(int kind, int msg, int timeArg, int from, int[] ids, int delay, int seed) {
    if (ops == null || (nops + 1) * 10 > ops.length) {
      int[] g = new int[Math.max(640, ops == null ? 0 : ops.length * 2)];
      if (ops != null) System.arraycopy(ops, 0, g, 0, nops * 10);
      ops = g;
    }
    int o = nops * 10;
    ops[o] = cur;
    ops[o + 1] = kind;
    ops[o + 2] = msg;
    ops[o + 3] = 0;
    ops[o + 4] = timeArg;
    ops[o + 5] = from;
    ops[o + 7] = ids.length;
    ops[o + 8] = delay;
    ops[o + 9] = seed;
    if (ids.length == 1) {
      ops[o + 6] = ids[0];
    } else {
      ops[o + 6] = nOpDests;
      if (nOpDests + ids.length > opDests.length) opDests = java.util.Arrays.copyOf(opDests, Math.max(opDests.length * 2, nOpDests + ids.length));
      System.arraycopy(ids, 0, opDests, nOpDests, ids.length);
      nOpDests += ids.length;
    }
    nops++;
  }

  private boolean stepOpen = false;

  // ---- an init() that sends BETWEEN node constructions (P/Paxos.java:374-387: every ProposerNode starts its first proposal
  // before the next one is built). The engine takes its node table whole, so inside deferredInit rd stays Java's (the node
  // constructors' draws, the protocol's, the sends' seed draws, in the reference's order), the sends and tasks are kept, and
  // once the nodes are added they are issued in their order — each send with the engine's rd put to where its seed draw
  // (:377, :430) found Java's: the same envelopes, the same push order as the interleaved original.
  // (hostnet.HostNetwork.deferred_init is this method; tests/test_gpu_paxos.py runs it.)
  private ArrayList<Object[]> deferred = null;

  public void deferredInit(Runnable init) {
    if (ready) throw new IllegalStateException("deferredInit() after the first send / run");
    deferred = new ArrayList<>();
    ArrayList<Object[]> kept;
    try {
      init.run();
      kept = deferred;
    } finally {
      deferred = null;
    }
    start();
    for (final Object[] o : kept) {
      @SuppressWarnings("unchecked")
      Message<? extends TN> m = (Message<? extends TN>) o[1];
      switch ((Integer) o[0]) {
        case WittGpu.OP_SEND:
          WittGpu.rngSetState(handle, (Long) o[2]);
          underHandle(m, h -> WittGpu.send(handle, h, 0, (Integer) o[3], (Integer) o[4], (int[]) o[5], (Integer) o[6]));
          break;
        case WittGpu.OP_SEND_ARRIVE_AT:
          underHandle(m, h -> WittGpu.sendArriveAt(handle, h, 0, (Integer) o[3], (Integer) o[4], ((int[]) o[5])[0]));
          break;
        default:
          underHandle(m, h -> WittGpu.registerTask(handle, h, 0, (Integer) o[3], (Integer) o[4]));
      }
    }
    // The replay put the engine's rd to each kept send's own state; it must END where init() itself ended — Java's rd, which
    // went on drawing inside init() (after the last send, or with no send at all: tasks only). The first stepBegin copies the
    // engine's state into Java's rd, so a stale one here would replace the right stream (ADVICE.md round 5; the Python mirror
    // ends deferred_init with wg_rng_set_state(end): tests/test_gpu_hostmode.py::test_deferred_init_leaves_rd_where_init_left_it).
    rdToEngine();
  }

  @Override
  public void send(Message<? extends TN> mc, int sendTime, TN fromNode, TN toNode) { // :369-382
    checkFrom(fromNode);
    if (toNode.nodeId >= allNodes.size() || getNodeById(toNode.nodeId) != toNode)
      throw new IllegalArgumentException("The from node is not in the network. To=" + toNode);
    sendIds(mc, sendTime, fromNode, new int[] {toNode.nodeId}, 0);
  }

  @Override
  public void send(Message<? extends TN> m, int sendTime, TN fromNode, List<? extends Node> dests, int delaysBetweenMessage) { // :420-447
    checkFrom(fromNode);
    int[] ids = new int[dests.size()];
    for (int i = 0; i < ids.length; i++) ids[i] = dests.get(i).nodeId;
    sendIds(m, sendTime, fromNode, ids, delaysBetweenMessage);
  }

  private void sendIds(Message<? extends TN> m, int sendTime, TN fromNode, int[] ids, int delay) {
    if (deferred == null) start();
    // createMessageArrival counts the sender's statistics for every destination, dropped or not (:476-477)
    GpuNodeAccess.sent(fromNode, ids.length, (long) ids.length * m.size());
    if (deferred != null) { // (deferredInit: the seed draw is made now, the send itself once the nodes are added)
      long before = stateOf(rd);
      rd.nextInt();
      deferred.add(new Object[] {WittGpu.OP_SEND, m, before, sendTime, fromNode.nodeId, ids, delay});
      return;
    }
    if (stepOpen) { // inside a delivery's action(): the seed draw (:377 / :430) is made HERE, in action() order
      int seed = rd.nextInt();
      if (ids.length > 0) op(WittGpu.OP_SEND, handleOf(m), sendTime, fromNode.nodeId, ids, delay, seed);
      return;
    }
    rdToEngine();
    underHandle(m, h -> WittGpu.send(handle, h, 0, sendTime, fromNode.nodeId, ids, delay)); // draws rd.nextInt(), also for an empty list (:430)
    rdFromEngine();
  }

  @Override
  public void sendArriveAt(Message<? extends TN> mc, int arriveAt, TN fromNode, TN toNode) { // :384-390
    if (deferred != null) {
      deferred.add(new Object[] {WittGpu.OP_SEND_ARRIVE_AT, mc, 0L, arriveAt, fromNode.nodeId, new int[] {toNode.nodeId}, 0});
      return;
    }
    start();
    if (arriveAt <= time) throw new IllegalArgumentException("wrong arrival time: arriveAt=" + arriveAt + ", time=" + time);
    if (stepOpen) op(WittGpu.OP_SEND_ARRIVE_AT, handleOf(mc), arriveAt, fromNode.nodeId, new int[] {toNode.nodeId}, 0, 0);
    else underHandle(mc, h -> WittGpu.sendArriveAt(handle, h, 0, arriveAt, fromNode.nodeId, toNode.nodeId));
  }

  private void register(Task<TN> t, int startAt, TN fromNode) {
    if (deferred != null) {
      deferred.add(new Object[] {WittGpu.OP_TASK, t, 0L, startAt, fromNode.nodeId, new int[0], 0});
      return;
    }
    start();
    if (stepOpen) op(WittGpu.OP_TASK, handleOf(t), startAt, fromNode.nodeId, new int[0], 0, 0);
    else underHandle(t, h -> WittGpu.registerTask(handle, h, 0, startAt, fromNode.nodeId));
  }

  @Override
  public void registerTask(final Runnable task, int startAt, TN fromNode) { // :505-508
    register(new Task<>(task), startAt, fromNode);
  }

  @Override
  public void registerPeriodicTask(final Runnable task, int startAt, int period, TN fromNode) { // :510-513
    register(new PeriodicTask<>(task, fromNode, period), startAt, fromNode);
  }

  @Override
  public void registerPeriodicTask(final Runnable task, int startAt, int period, TN fromNode, Condition c) { // :515-519
    register(new PeriodicTask<>(task, fromNode, period, c), startAt, fromNode);
  }

  @Override
  public void registerConditionalTask(final Runnable task, int startAt, int duration, TN fromNode, Condition startIf, Condition repeatIf) { // :521-531
    condTasks.add(new ConditionalTask<>(startIf, repeatIf, task, startAt, fromNode, duration));
  }

  @Override
  public boolean hasMessage() { // msgs.size() != 0
    start();
    return WittGpu.queueSize(handle) != 0;
  }

  /** msgs.size() / msgs.sizeAt(t) (C/Network.java:204-220; `msgs` itself is package-private in the reference) */
  public long queueSize() {
    start();
    return WittGpu.queueSize(handle);
  }

  public long queueSizeAt(int t) {
    start();
    return WittGpu.queueSizeAt(handle, t);
  }

  @Override
  public void partition(float part) { // :693-703 (the checks stay the superclass's)
    super.partition(part);
    start();
    WittGpu.setPartitions(handle, cuts());
  }

  @Override
  public void endPartition() {
    super.endPartition();
    if (ready) WittGpu.setPartitions(handle, new int[0]);
  }

  private int[] cuts() {
    List<Integer> px = GpuNodeAccess.partitionsInX(this);
    int[] c = new int[px.size()];
    for (int i = 0; i < c.length; i++) c[i] = px.get(i);
    return c;
  }

  // ---- the loop
  @Override
  public boolean runMs(int ms) { // :318-338
    if (ms <= 0) throw new IllegalArgumentException("Should be greater than 0. ms=" + ms);
    start();
    if (resident) { // action() runs on the device: one native call
      rdToEngine();
      boolean did = WittGpu.runMs(handle, ms, null);
      rdFromEngine();
      time = WittGpu.time(handle);
      return did;
    }
    if (time == 0) for (Node n : allNodes) if (!n.isDown()) n.start();
    int endAt = time + ms;
    if (endAt <= 0) throw new IllegalStateException("Maximum time reached!");
    drainReleased(); // (sends of init() that reached no destination)
    boolean did = receiveUntilGpu(endAt);
    drainReleased(); // (envelopes whose last hop was consumed, not delivered: no delivery showed their end)
    time = endAt;
    WittGpu.setTime(handle, endAt);
    return did;
  }

  private int condTime(List<ConditionalTask<TN>> cts, int until) { // the earliest minStartTime the edge scan would act on
    int t = Integer.MAX_VALUE;
    for (ConditionalTask<TN> ct : cts == null ? condTasks : cts)
      if (ct.minStartTime <= until && !ct.from.isDown()) t = Math.min(t, ct.minStartTime);
    return t;
  }

  /** the conditional-task scan of a time++ edge (C/Network.java:543-566) */
  private List<ConditionalTask<TN>> edge(List<ConditionalTask<TN>> cts, int until) {
    if (cts == null) cts = new ArrayList<>(condTasks);
    Iterator<ConditionalTask<TN>> it = cts.iterator();
    while (it.hasNext()) {
      ConditionalTask<TN> ct = it.next();
      if (ct.minStartTime > until || ct.from.isDown()) {
        it.remove();
        continue;
      }
      if (ct.minStartTime <= time) {
        it.remove();
        if (ct.startIf.check()) {
          ct.r.run();
          ct.minStartTime = time + ct.duration;
          if (!ct.repeatIf.check()) condTasks.remove(ct);
        }
      }
    }
    return cts;
  }

  /** receiveUntil (C/Network.java:587-637) + nextMessage (:533-570) over wg_step_begin / wg_step_end */
  @SuppressWarnings("unchecked")
  private boolean receiveUntilGpu(int until) {
    boolean did = false;
    List<ConditionalTask<TN>> cts = null; // nextMessage()'s private copy, made at the first edge of a call
    int n;
    while ((n = WittGpu.stepBegin(handle, until, condTime(cts, until), batch)) > 0) {
      // rd is this class's while the step is open: send() draws its seed inside action(), in order with the action()'s own draws
      setState(rd, WittGpu.rngGetState(handle));
      nops = 0;
      nOpDests = 0;
      stepOpen = true;
      RuntimeException failed = null;
      try {
        for (int i = 0; i < n; i++) {
          int o = 6 * i;
          cur = i;
          time = batch[o + 1];
          if (batch[o] == 2) { // a time edge
            cts = edge(cts, until);
            continue;
          }
          did = true;
          cts = null; // a delivery ends the nextMessage() call
          TN from = allNodes.get(batch[o + 2]), to = allNodes.get(batch[o + 3]);
          if (to.isDown()) continue; // stopped by an earlier action() of this very step (:606)
          Message<TN> m = (Message<TN>) byHandle.get(batch[o + 4]);
          if (!(m instanceof Task<?>)) { // :607-613
            if (m.size() == 0) throw new IllegalStateException("Message size should be greater than zero: " + m);
            GpuNodeAccess.received(to, m.size());
          }
          m.action(this, from, to); // :616-626 — may call send() / registerTask(): recorded as ops
        }
      } catch (RuntimeException x) {
        failed = x; // the step is still closed below, with the pushes made so far
      } finally {
        stepOpen = false;
        WittGpu.rngSetState(handle, stateOf(rd));
      }
      try {
        WittGpu.stepEnd(handle, ops == null ? new int[0] : ops, nops, opDests);
      } catch (RuntimeException refused) {
        // wg_step_end refuses a step as a whole and leaves it open (nothing applied): close it without the action()s' pushes —
        // the multi-destination envelopes' owed re-pushes are kept —, forget the envelopes that were never made, report
        WittGpu.stepEnd(handle, new int[0], 0, opDests);
        for (int k = 0; k < nops; k++) unref(ops[10 * k + 2]);
        drainReleased();
        throw failed != null ? failed : refused;
      }
      drainReleased();
      if (failed != null) throw failed;
    }
    return did;
  }
}
