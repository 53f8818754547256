package net.consensys.wittgenstein.core.gpu;

import java.nio.IntBuffer;

/**
 * Static binding of libwittgpu.so: one native method per export of include/wittgpu.h and include/wittgpu_host.h
 * (jni/wittgpu_jni.c, same order). An engine / batch is a {@code long} handle. A non-zero status arrives as the
 * exception the reference throws at the corresponding site — IllegalArgumentException (WG_EINVAL: C/Network.java:320,
 * 371,374,386,427,695), IllegalStateException (WG_ESTATE: :137,250,333,472,599,609,656,671; also WG_ENOMEM / WG_EHIP with
 * the engine's text), UnsupportedOperationException (WG_EUNSUPPORTED); {@link #EHOSTINIT} is returned, not thrown.
 *
 * <p>Not compiled in this repository's build image (no JDK there); the C ABI underneath is what its tests exercise.
 */
public final class WittGpu {
  static {
    System.loadLibrary("wittgpu_jni"); // JNI_OnLoad checks wg_abi_version() and the struct sizes
  }

  private WittGpu() {}

  public static final int OK = 0, EINVAL = -1, ESTATE = -2, ENOMEM = -3, EHIP = -4, EUNSUPPORTED = -5, EHOSTINIT = -6;
  /** wg_latency_kind */
  public static final int LAT_BY_DISTANCE_WJITTER = 0, LAT_FIXED = 1, LAT_UNIFORM = 2, LAT_NONE = 3, LAT_MEASURED = 4,
      LAT_IC3 = 5, LAT_ETHSCAN = 6;
  /** wg_field (wg_read_i64 / wg_read_i32) */
  public static final int F_DONE_AT = 0, F_MSG_RECEIVED = 1, F_MSG_SENT = 2, F_BYTES_SENT = 3, F_BYTES_RECEIVED = 4,
      F_DOWN = 5, F_X = 6, F_Y = 7, F_EXTRA_LATENCY = 8, F_PONG = 16, F_SIGS_CHECKED = 32, F_SIG_QUEUE_SIZE = 33,
      F_MSG_FILTERED = 34, F_CURR_WINDOW_SIZE = 35, F_ADDED_CYCLE = 36, F_START_AT = 37, F_NODE_PAIRING_TIME = 38,
      F_GSF_SIG_CHECKED = 48, F_GSF_SIG_QUEUE_SIZE = 49, F_GSF_TO_VERIFY_SIZE = 50, F_GSF_VERIFIED_CARDINALITY = 51,
      F_SF_AGG_VALUE = 64, F_SF_PREFIX_LENGTH = 65, F_SF_FLAGS = 66, F_SF_SENT_REQUESTS = 67, F_SF_RECEIVED_REQUESTS = 68,
      F_SF_THRESHOLD_AT = 69, F_CASPER_HEAD_HEIGHT = 80, F_CASPER_HEAD_TIME = 81, F_CASPER_HEAD_ID = 82,
      F_CASPER_HEADS_ATTESTED = 83, F_CASPER_BLOCKS_RECEIVED = 84, F_CASPER_ATTESTATIONS_HELD = 85, F_FLOOD_RECEIVED = 96,
      F_FLOOD_PEER_COUNT = 97;
  /** wg_level_field */
  public static final int LF_POS_IN_LEVEL = 0, LF_OUTGOING_FINISHED = 1, LF_QUEUE_LEN = 2, LF_REMAINING_CALLS = 3,
      LF_RECEPTION_RANKS = 4, LF_SUICIDE_BIZ_AFTER = 5;
  /** wg_bits_field */
  public static final int B_TOTAL_INCOMING = 0, B_LAST_AGG_VERIFIED = 1, B_VERIFIED_IND = 2, B_TO_VERIFY_IND = 3,
      B_FINISHED_PEERS = 4, B_BLACKLIST = 5, B_GSF_VERIFIED = 8, B_GSF_INDIVIDUAL = 9, B_GSF_INDIV_VERIFIED = 10;
  /** wg_step_op_kind */
  public static final int OP_SEND = 0, OP_SEND_ARRIVE_AT = 1, OP_TASK = 2;

  public static native int abiVersion();

  public static native int abiStructSize(int which);

  /** wg_selftest: one wave / block primitive of the device kernels run on the caller's values (include/wittgpu.h) */
  public static native int selfTest(int op, int aux, long[] in, int n, int threads, long[] out);

  // ---- lifecycle. cfgInts = {device, horizon_ms, queue_cap, queue_cap_wide, chain_slots, shard, nshards, rank_bump_cap} (null / shorter:
  // zeros = defaults), cfgLongs = {bucket_pool_records, payload_words, outbox_records, chain_dests}, rcclId = null or 128 bytes
  public static native long create(int[] cfgInts, long[] cfgLongs, byte[] rcclId);

  public static native void destroy(long h);

  public static native String lastError(long h);

  // ---- topology
  public static native int addNodes(long h, int[] x, int[] y, int[] extraLatency, byte[] down, byte[] byzantine, double[] speedRatio);

  public static native int nodeCount(long h);

  public static native int setLatency(long h, int kind, int[] params);

  public static native int setLatencyCity(long h, int mode, int nCities, int[] cityOfNode, int[] tab, float[] ping, double[] jitter100);

  public static native int setLatencyByName(long h, String name);

  public static native int latencyProbe(long h, int[] from, int[] to, int[] delta, int[] out);

  public static native int setPartitions(long h, int[] xcuts);

  public static native int setNodeDown(long h, int id, boolean down);

  public static native int setDiscardTime(long h, int ms);

  // ---- rd
  public static native int rngSetSeed(long h, long seed);

  public static native long rngGetState(long h);

  public static native int rngSetState(long h, long s48);

  // ---- host-side sends / tasks
  public static native int send(long h, int msg, int payload, int sendTime, int from, int[] dests, int delayBetween);

  public static native int sendArriveAt(long h, int msg, int payload, int arriveAt, int from, int to);

  public static native int registerTask(long h, int task, int arg, int startAt, int node);

  public static native int registerPeriodicTask(long h, int task, int startAt, int period, int node);

  // ---- resident protocols (params: the int fields of the wg_*_params struct, in order)
  public static native int loadHost(long h);

  public static native int loadPingPong(long h);

  public static native int loadHandel(long h, int[] params14, int[] startAt, int[] nodePairingTime, IntBuffer receptionRanks, IntBuffer peers);

  public static native int loadGsf(long h, int[] params7, int[] nodePairingTime, IntBuffer peers);

  public static native int loadSanFermin(long h, int[] params6);

  public static native int loadCasper(long h, int[] params8);

  public static native int loadP2PFlood(long h, int[] params7, int[] peers, int[] peerCount, int maxPeers, int[] senders);

  // ---- run. stats7 (nullable) = {delivered, tasks, events, draws, simulated_ms, wall_ns, payload_bytes}
  public static native boolean runMs(long h, int ms, long[] stats7);

  public static native int time(long h);

  public static native long queueSize(long h);

  public static native long queueSizeAt(long h, int t);

  public static native boolean protocolContIf(long h);

  // ---- the init() image
  public static native int snapshot(long h);

  public static native int restore(long h);

  public static native long snapshotBytes(long h);

  // ---- host-callback mode. A delivery is six ints {kind, time, from, to, msg, payload}; an op ten ints in wg_step_op's order
  public static native boolean nextDelivery(long h, int until, int condTime, int[] out6);

  public static native int setTime(long h, int time);

  public static native int stepBegin(long h, int until, int condTime, int[] batch6);

  public static native int stepEnd(long h, int[] ops10, int nops, int[] dests);

  /** wg_host_released: handles whose envelope ended since the last call, oldest first, at most out.length; returns how many */
  public static native int hostReleased(long h, int[] out);

  // ---- batches
  public static native long batchCreate(long[] handles);

  public static native void batchDestroy(long batch);

  public static native String batchLastError(long batch);

  /** members of the batch: the length every per-member array of the calls below must have */
  public static native int batchSize(long batch);

  public static native int batchRunMs(long batch, int ms, byte[] active, byte[] didSomething, long[] stats7n);

  public static native int batchContIf(long batch, int[] cont);

  public static native int batchRunMultipleTimes(long batch, int chunk, int maxTime, long[] delivered, long[] simulatedMs);

  // ---- node-range sharding
  public static native byte[] rcclUniqueId();

  public static native int shardConfigureRccl(long h, int shard, int nshards, byte[] id128);

  public static native int shardConfigure(long h, int shard, int nshards, long allreduceFnAddress, long ctxAddress);

  /** wg_shard_set_alltoallv: the address of a native wg_alltoallv_fn and its context (after shardConfigure, before protocolLoad) */
  public static native int shardSetAlltoallv(long h, long alltoallvFnAddress, long ctxAddress);
  public static native int shardInfo(long h, long[] loHiCollectivesWords);
  /** wg_shard_traffic: collective calls / int32 words by exchange (eight entries each) */
  public static native int shardTraffic(long h, long[] calls8, long[] words8);

  // ---- read-back
  public static native int readI64(long h, int field, long[] dst);

  public static native int readI32(long h, int field, int[] dst);

  public static native int readLevelI32(long h, int field, int[] dst, int nNodes, int nLevels);

  public static native int readBits(long h, int field, long[] dst, int nNodes, int wordsPerNode);

  public static native int levels(long h);

  public static native long deviceBytes(long h);

  public static native int deliveredByLevel(long h, long[] dst32);

  // ---- measurement
  public static native int profileEnable(long h, int mode);

  public static native int profileRead(long h, String[] names, long[] spans, double[] totalNs);

  public static native int profileSetReference(long h, long ref);

  public static native int profileReadSpans(long h, int cls, double[] startNs, double[] endNs);

  // ---- include/wittgpu_host.h: the engine's own restatements of Protocol.init()
  public static native long hostPingPongCreate(int nodeCt, String nodeBuilderName, String latencyName, long seed, int[] cfgInts, long[] cfgLongs, byte[] rcclId);

  public static native long hostHandelCreate(int[] params14, String nodeBuilderName, String latencyName, long seed, int[] cfgInts, long[] cfgLongs, byte[] rcclId);

  public static native long hostHandelCreateBadNodes(int[] params14, byte[] badNodes, String nodeBuilderName, String latencyName, long seed, int[] cfgInts, long[] cfgLongs, byte[] rcclId);

  public static native long hostGsfCreate(int[] params7, String nodeBuilderName, String latencyName, long seed, int[] cfgInts, long[] cfgLongs, byte[] rcclId);

  public static native long hostSanFerminCreate(int[] params6, String nodeBuilderName, String latencyName, long seed, int[] cfgInts, long[] cfgLongs, byte[] rcclId);

  public static native long hostCasperCreate(int[] params8, String nodeBuilderName, String latencyName, long seed, int[] cfgInts, long[] cfgLongs, byte[] rcclId);

  public static native long hostP2PFloodCreate(int[] params7, String nodeBuilderName, String latencyName, long seed, int[] cfgInts, long[] cfgLongs, byte[] rcclId);

  public static native int registerCityBuilder(String site, float[] cumulativeProbability, int[] mercX, int[] mercY, int listSize);

  public static native int registerCityLatency(String latencyName, int mode, int nCities, int[] tab, float[] ping, double[] jitter100);

  public static native String hostLastError();

  public static native double hostLastInitSeconds();

  public static native boolean hostLastInitOnDevice();

  public static native int jrandomInts(long seed, int[] out);

  public static native int jrandomSkipInts(long seed, int[] out);

  public static native int jrandomBounded(long seed, int bound, int[] out);
}
