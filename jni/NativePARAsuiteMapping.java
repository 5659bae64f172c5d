/*
 * NativePARAsuiteMapping -- JNI route into libparasuite_hip.so (the MI355X replacement of the `bwa index` /
 * `bwa parasuite|aln` / `bwa samse` child processes).  Drop-in subclass of mapping.Mapping
 * (/root/reference/src/src/mapping/Mapping.java:40-42) with PARAsuiteMapping's setters
 * (PARAsuiteMapping.java:22-28); Main.java:316 would instantiate it instead of PARAsuiteMapping.
 *
 * NOT compiled or exercised in this repository's image (no JDK, no jni.h): source a maintainer adds to
 * src/src/mapping/ of the reference.  The C ABI it calls is include/parasuite_hip.h.
 */
package mapping;

public class NativePARAsuiteMapping extends Mapping {
    static { System.loadLibrary("parasuite_jni"); }          // libparasuite_jni.so next to libparasuite_hip.so

    private String errorProfileFilename, indelProfileFilename;
    public void setErrorProfileFilename(String f) { errorProfileFilename = f; }   // PARAsuiteMapping.java:22-24
    public void setIndelProfileFilename(String f) { indelProfileFilename = f; }   // PARAsuiteMapping.java:26-28

    private static native int nativeIndex(String reference);
    private static native int nativeMap(int threads, String mm, String errorProfile, String indelProfile,
                                        String reference, String input, String outSam);
    /** first pass + the error profile of its alignments with MAPQ >= minMapq in one call (instead of Main.java:288-334) */
    public static native int nativeMapProfiled(int threads, String mm, String reference, String input, String outSam,
                                               int minMapq, int maxReadLength, String profilePrefix);
    /** ErrorProfiling.inferErrorProfile(false, false) on an existing SAM / BAM file (Main.java:327-334) */
    public static native int nativeErrorProfile(String mapping, String reference, int maxReadLength, String outPrefix);
    private static native String nativeLastError();

    /* errorProfileFilename == null: the stock first pass (BWAMapping.java:51-75, `bwa aln -n mm`) */
    public void executeMapping(int threads, String reference, String input, String outputPrefix,
                               int mappingQualityFilter, String additionalOptions) {
        if (!new java.io.File(reference + ".bwt").exists() && nativeIndex(reference) != 0)   // PARAsuiteMapping.java:45-55
            fail("ps_index " + reference);
        setTimeStart();                                                                       // :57
        if (nativeMap(threads, additionalOptions, errorProfileFilename, indelProfileFilename,
                      reference, input, outputPrefix + ".sam") != 0)                          // :63-92
            fail("ps_map " + reference + " " + input);
        // from here on unchanged: samtools view -bS / view -q / rm / mv  (PARAsuiteMapping.java:102-152),
        // or one call of ps_sam_to_bam (INTEGRATION.md section D)
    }

    private void fail(String what) {          // same contract as Mapping.executeCommand, Mapping.java:170-197
        main.MappingLogger.getLogger().error("External program had non-zero exit status: " + what + ": " + nativeLastError());
        System.exit(1);
    }
}
